#!/usr/bin/env python3
"""How many code lines of the Python mirror are character-identical to lines of the reference?
(the mirror had been written with the reference open; the bar is < 10 % identical code
lines, signatures and exception strings aside.)  Needs /root/reference; CPU only.

usage: tools/copycheck.py [--show] file.py ..."""
import glob
import io
import sys
import tokenize

REF = "/root/reference/pygsp"
TRIVIAL = {"else:", "try:", "return", "pass", "continue", "break", ")", "(", "]", "[", "}", "{", "finally:"}


def code_lines(path):
    """Stripped source lines that carry code: comments, blank lines and docstrings removed."""
    src = open(path).read()
    drop = set()
    prev = None
    for tok in tokenize.generate_tokens(io.StringIO(src).readline):
        if tok.type == tokenize.STRING and (prev is None or prev.type in (tokenize.NEWLINE, tokenize.INDENT,
                                                                         tokenize.DEDENT, tokenize.NL)):
            drop.update(range(tok.start[0], tok.end[0] + 1))  # a docstring / bare string statement
        if tok.type not in (tokenize.COMMENT, tokenize.NL):
            prev = tok
    out = []
    for i, line in enumerate(src.split("\n"), 1):
        t = line.strip()
        if not t or t.startswith("#") or i in drop:
            continue
        out.append(t)
    return out


def main():
    show = "--show" in sys.argv
    files = [a for a in sys.argv[1:] if not a.startswith("--")]
    ref = set()
    for f in glob.glob(REF + "/**/*.py", recursive=True):
        if "/tests/" not in f:
            ref.update(code_lines(f))
    for f in files:
        cl = code_lines(f)
        same = [t for t in cl if t in ref and t not in TRIVIAL]
        # the interface a drop-in must share with the reference: signatures, decorators, imports, and the
        # exception messages callers match on
        contract = ("def ", "class ", "@", "import ", "from ", "raise ", "except ")
        body = [t for t in same if not t.startswith(contract)]
        print("{:32s} code lines {:4d}  identical to a reference line {:4d} = {:4.1f} %   "
              "of which not signature / import / raise: {:4d} = {:4.1f} %".format(
                  f, len(cl), len(same), 100.0 * len(same) / max(len(cl), 1), len(body),
                  100.0 * len(body) / max(len(cl), 1)))
        same = body if "--body" in sys.argv else same
        if show:
            for t in same:
                print("      " + t)


if __name__ == "__main__":
    main()
