"""config 5 loop (examples/inverted_pendulum.run_hip) with the RbfController inline (default) and as launches of its own."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
from pilco_amd import _lib
import inverted_pendulum as ip
ctx = _lib.get_context()
for mode in (1, 0, 1, 0):
    ctx.set_inline_policy(mode)
    r = ip.run_hip(verbose=False)
    print("inline_policy=%d: total %.3f s; optimize_policy per iteration %s; optimize_models %s" % (
        mode, r["total_s"], ["%.3f" % i["optimize_policy_s"] for i in r["iterations"]], ["%.3f" % i["optimize_models_s"] for i in r["iterations"]]), flush=True)
