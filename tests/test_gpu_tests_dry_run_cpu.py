"""A dry run of part of the GPU parity suite on the box WITHOUT a GPU: the selected tests of tests/test_gpu_parity.py are run
in a child pytest whose default context is the oracle stand-in (tests/helpers/standin_plugin.py).  Nothing of the device is
checked here -- that is what `-m gpu` on an MI355X does -- but the tests' own logic, the fixtures they load and the product's
whole Python layer under them are, every round, so that a change of the host code or of a test cannot first fail on the GPU
box.  (The child run selects gpu-marked tests on purpose; they need no GPU in this mode.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECT = ("predictions_golden or cascade_golden or cascade_trained or controllers_and_reward or rbf_controller_golden or "
          "mgpr_optimize_ends or smgpr_optimize_ends or optimize_policy_ends or optimize_policy_rbf_ends or host_reward_terms or "
          "safe_pilco_vs_executed or safe_pilco_rbf or zero_covariance or two_models_share or degenerate_dims or "
          "sparse_rollout_and_policy or policy_gradients_vs_reverse")


def test_selected_gpu_parity_tests_pass_on_the_oracle_stand_in():
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-p", "helpers.standin_plugin", "-q",
           "-k", SELECT, "-p", "no:cacheprovider"]
    pr = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    tail = pr.stdout[-2500:]
    assert pr.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail and int(tail.split(" passed")[0].split()[-1]) >= 23, tail
