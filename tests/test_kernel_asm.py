"""Static check of the compiled gfx950 code of k_step_tile (no GPU needed): the partial wait before the workgroup
barrier - `s_waitcnt vmcnt(N) lgkmcnt(0)`, gspx_tile_kernels.hip.h - lets the N youngest vector-memory operations
stay in flight and must still cover every tile DMA (`buffer_load ... lds`, tracked by vmcnt only).  That holds
when the N vector-memory instructions issued immediately before the wait, in the same basic block, are ordinary
loads: everything older, the tile DMA included, has then completed.  A compiler that merges, scalarises or
hoists one of those loads would leave fewer than N younger operations and the wait would silently stop covering
the DMA: this test fails the build instead."""
import os
import re
import shutil
import subprocess

import pytest

from pygsp_amd import _capi

LLVM = "/opt/rocm/lib/llvm/bin"
VMEM = re.compile(r"^(buffer_load|buffer_store|buffer_atomic|global_load|global_store|global_atomic|flat_load|flat_store|"
                  r"flat_atomic|scratch_load|scratch_store)")


def _device_asm(tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin",
                           _capi.LIB_PATH, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--symbolize-operands", co], check=True,
                          capture_output=True, text=True).stdout


def _vmem_ops_certainly_after_the_last_dma(lines, w):
    """Lower bound, over every path that reaches the wait at line w, of the vector-memory operations issued after
    the textually last tile DMA above it: instructions inside a region that a forward branch can skip do not
    count; a label that can be entered from outside the region (a loop header reached over its back edge, a join
    with code before the DMA) restarts the count."""
    labels = {m.group(1): i for i, ln in enumerate(lines) for m in [re.match(r"<(\S+)>:$", ln)] if m}
    branches = [(i, m.group(2)) for i, ln in enumerate(lines)
                for m in [re.match(r"(s_cbranch\w*|s_branch)\s+(\S+)", ln)] if m]
    dma = [i for i in range(w) if lines[i].startswith("buffer_load") and " lds" in lines[i]]
    assert dma, "no tile DMA above the wait"
    s = dma[-1]
    count, skipping = 0, set()
    for i in range(s + 1, w):
        ln = lines[i]
        m = re.match(r"<(\S+)>:$", ln)
        if m:
            skipping.discard(m.group(1))
            if any(tgt == m.group(1) and (src < s or src > w) for src, tgt in branches):
                count, skipping = 0, set()
            continue
        m = re.match(r"(s_cbranch\w*|s_branch)\s+(\S+)", ln)
        if m:
            t = labels.get(m.group(2))
            if t is not None and i < t <= w:
                skipping.add(m.group(2))
            continue
        if VMEM.match(ln) and not skipping:
            count += 1
    return count


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")) or not shutil.which("c++filt"),
                    reason="needs the ROCm LLVM tools")
def test_partial_vmcnt_wait_of_the_tile_kernel_covers_the_tile_dma(tmp_path):
    asm = _device_asm(str(tmp_path))
    functions = re.split(r"\n(?=[0-9a-f]+ <_Z)", asm)
    checked = waits = 0
    for body in functions:
        head = re.match(r"[0-9a-f]+ <(\S+)>:", body)
        if not head or "k_step_tile" not in head.group(1):
            continue
        name = subprocess.run(["c++filt", head.group(1)], capture_output=True, text=True).stdout.strip()
        lines = [re.sub(r"\s*//.*", "", ln).strip() for ln in body.split("\n")[1:]]
        lines = [ln for ln in lines if ln]
        checked += 1
        assert any(" lds" in ln and ln.startswith("buffer_load") for ln in lines), name + ": no tile DMA found"
        for i, ln in enumerate(lines):
            m = re.match(r"s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)$", ln)
            if not m or int(m.group(1)) == 0 or not lines[i + 1].startswith("s_barrier"):
                continue
            waits += 1
            need, have = int(m.group(1)), _vmem_ops_certainly_after_the_last_dma(lines, i)
            assert have >= need, ("{}: vmcnt({}) before the workgroup barrier, but only {} vector-memory operations are "
                                  "certain to be younger than the tile DMA: the wait may leave a tile row in flight"
                                  ).format(name, need, have)
    assert checked >= 12, "k_step_tile builds not found in the device code"
    assert waits >= 4, "the partial wait was not found in any build (kernel changed? update this check)"
