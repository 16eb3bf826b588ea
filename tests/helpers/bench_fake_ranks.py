"""Run bench.py's control flow (argument handling, gloo rendezvous, id broadcast, barriers, max-over-ranks timing, the
replica leg, verification, the JSON line) with the device replaced by a stand-in Context, one process per rank on the
CPU.  TEST HELPER (tests/test_sharding_cpu.py launches it under torch.distributed.run): the multi-rank branch of
bench.py cannot run on the single-GPU test box, so its host logic is exercised here."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from pilco_amd import _lib  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "c2_rollout.npz"))


class FakeContext:
    """The slice of pilco_amd._lib.Context bench.py touches; a rollout returns the fixture's answer after a sleep."""
    calls = []

    def __init__(self, device=None):
        self.device = device
        self.rank, self.nranks = 0, 1

    def comm_unique_id(self):
        return bytes(range(128))

    def comm_init(self, id_bytes, rank, nranks):
        assert bytes(id_bytes) == bytes(range(128)), "rank 0's id was not broadcast intact"
        self.rank, self.nranks = rank, nranks
        FakeContext.calls.append(("comm_init", rank, nranks))

    # peer exchange: FAKE_PEER=ok attaches on every rank; FAKE_PEER=rank1_fails lets rank 1's attach fail (then every
    # rank must detach and stay on the RCCL path); FAKE_PEER=first_rollout_fails lets rank 1's first rollout time out
    attached = False

    def comm_count(self):
        return self.nranks

    def peer_export(self):
        return bytes([self.rank]) * 64

    def peer_attach(self, handles, share_gpu=False):
        assert [h[0] for h in handles] == list(range(self.nranks)), "handles must arrive in rank order"
        if os.environ.get("FAKE_PEER") == "rank1_fails" and self.rank == 1:
            raise _lib.PilcoError(3, "hipIpcOpenMemHandle: stand-in failure")
        self.attached = True

    def peer_detach(self):
        self.attached = False
        FakeContext.calls.append(("peer_detach", self.rank))

    def peer_attached(self):
        return self.attached

    def gp_set_data(self, slot, X, Y):
        assert X.shape == (1000, 10) and Y.shape == (1000, 10)

    def gp_set_hyp(self, slot, ls, var, nz):
        pass

    def gp_set_inducing(self, slot, Z):
        pass

    def gp_factorize(self, slot):
        FakeContext.calls.append(("factorize", self.rank, self.nranks))

    def rollout(self, policy, rewards, m0, S0, H, want_traj=False):
        if self.attached and os.environ.get("FAKE_PEER") == "first_rollout_fails" and self.rank == 1:
            raise _lib.PilcoError(5, "rollout: peer exchange 1 timed out on rank 1 (stand-in)")
        time.sleep(0.002 * (1 + self.rank))            # ranks differ: the reported time must be the slowest one's
        return G["M_traj"][:, -1][None, :].copy(), G["S_traj"][:, :, -1].copy(), np.array([[G["R_traj"][-1]]])

    def rollout_timed(self, policy, rewards, m0, S0, H, reps, time_pair=True):
        return dict(ms_total=2.6 * reps, ms_pair=0.047 * H, n_pair_launches=H)

    def close(self):
        pass


_lib.Context = FakeContext
bench.engine_clock_under_pair_load = lambda *a, **k: (2000.0, 45.0)

if __name__ == "__main__":
    bench.main()
