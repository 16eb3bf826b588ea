"""config 5 loop with hipGraph replay off (eager launches) -- for kernel traces (rocprofv3 + graphs of this size crashed)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
from pilco_amd import _lib
ctx = _lib.get_context()
ctx.use_graph(False)
import inverted_pendulum as ip
r = ip.run_hip(verbose=True)
print("total %.2f s" % r["total_s"])
