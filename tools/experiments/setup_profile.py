import sys, time; sys.path.insert(0,".")
import numpy as np, cProfile, pstats
from pygsp_amd import graphs, engine
W,c=graphs.sensor_weights(1000000,k=8,seed=42)
engine.default_context(0)
G0=graphs.Graph(W[:5000,:5000],coords=c[:5000])
pr=cProfile.Profile(); pr.enable()
G=graphs.Graph(W,coords=c)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
