"""bench.py on another build of the library (an A/B of two kernels on one GPU box):
    python tools/bench_with.py tools/diag/libspicegpu_variant.so [bench.py's arguments]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import spice_amd
spice_amd.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
