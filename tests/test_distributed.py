"""Multi-rank path on CPU: world_size-2 (and 3) gloo runs of the sharded iteration driver
(sparseharness_amd/distributed.py) with the oracle as the local step, checked bit-for-bit
against the single-process oracle loop.  What is under test is the product's sharding,
slotted layout, column remap, all-gather and termination logic -- the device kernel is
covered by the -m gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden
from oracle import oracle as O
from sparseharness_amd import hostlib as H
from sparseharness_amd import partition
from sparseharness_amd.distributed import ShardedIteration, ShardPlan


class OracleLocalStep:
    """Test double for HipLocalStep: same contract, computed by the CPU oracle."""

    device = torch.device("cpu")

    def __init__(self, plan, semiring):
        self.plan, self.semiring = plan, semiring

    def launch(self, x_cur, y_vec, x_next, alpha, beta, delta):
        """All pieces of the rank (the HIP step is one launch that reports its pieces one by one; this double computes
        them one by one up front)."""
        for c in range(self.plan.chunks):
            off = self.plan.layout.piece_offset(self.plan.rank, c)
            rows = self.plan.pieces[c][3]
            self.step(c, x_cur, y_vec[off:off + max(rows, 1)], x_next, alpha, beta, delta)

    def wait_piece(self, c):
        pass

    def step(self, c, x_cur, y_piece, x_next, alpha, beta, delta):
        p, lay = self.plan, self.plan.layout
        rp, ci, va, rows = p.pieces[c]
        if rows == 0:
            return
        dt = O.elem_dtype(self.semiring)
        x = x_cur.numpy().view(dt)
        y = y_piece.numpy().view(dt)[:rows]
        out = O.kernel(self.semiring, rp, ci, va.astype(dt), x, y, alpha, beta, vlength=lay.length)
        off = lay.piece_offset(p.rank, c)
        prev = x[off:off + rows]
        if self.semiring in (O.OR_AND_I32, O.MAX_MIN_I32):
            changed = bool((prev != out).any())
        else:
            changed = bool((~(np.abs(prev - out).astype(np.float64) < delta)).any())
        xn = x_next.numpy().view(dt)
        xn[off:off + rows] = out
        if changed:   # (the driver cleared the word before the launch; every piece may raise it)
            x_next.numpy().view(np.int32)[lay.flag_index(p.rank)] = 1


def initial_y(sr, x0):
    """y0 of the app: x0 for sssp/bfs, INT_MIN everywhere for scc (app/scc.cpp:201-202)."""
    return np.full(len(x0), O.INT_MIN, np.int32) if sr == O.MAX_MIN_I32 else x0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, q, chunks=1, exchange="collective"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sr, rp, ci, va, a, b = case
        n = len(rp) - 1
        plan = ShardPlan(rp, ci, va, rank, world, chunks)
        x0 = O.initial_vector(sr, n)
        final, iters, conv = ShardedIteration(plan, sr, OracleLocalStep(plan, sr), exchange=exchange).run(
            x0, initial_y(sr, x0), a, b, 1e-4, 500)
        q.put((rank, final, iters, conv))
    finally:
        dist.destroy_process_group()


def run_world(world, case, chunks=1, exchange="collective"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q, chunks, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda t: t[0])


def cases():
    out = {}
    rp, ci, va = H.rmat(11, seed=77)
    out["rmat11_sssp"] = (O.MIN_PLUS_F32, rp, ci, va, 0.0, 0.0)
    out["rmat11_bfs"] = (O.OR_AND_I32, rp, ci, va.astype(np.int32), 1, 0)
    g = golden("matrix")
    out["1138bus_sssp"] = (O.MIN_PLUS_F32, g["f32_row_ptr"], g["f32_col_idx"], g["f32_val"], 0.0, 0.0)
    out["1138bus_bfs"] = (O.OR_AND_I32, g["i32_row_ptr"], g["i32_col_idx"], g["i32_val"], 1, 0)
    g5 = golden("matrix5")   # SCC on the reference's rows after scc_normalise (3 launches, 20 labels)
    out["matrix5_scc"] = (O.MAX_MIN_I32, g5["scc_row_ptr"], g5["scc_col_idx"], g5["scc_val"], O.INT_MAX, O.INT_MIN)
    return out


@pytest.mark.parametrize("name", ["rmat11_sssp", "rmat11_bfs", "1138bus_sssp", "1138bus_bfs", "matrix5_scc"])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_iteration_matches_single_process(name, world):
    if world == 3 and not name.startswith(("1138bus", "matrix5")):
        pytest.skip("3 ranks exercised on one graph only (keeps the CPU suite short)")
    case = cases()[name]
    sr, rp, ci, va, a, b = case
    n = len(rp) - 1
    x0 = O.initial_vector(sr, n)
    want, w_it, w_conv = O.iterate(sr, rp, ci, va, x0, initial_y(sr, x0), a, b, 1e-4, 500)
    res = run_world(world, case)
    for rank, final, iters, conv in res:
        assert (iters, conv) == (w_it, w_conv), f"rank {rank}"
        np.testing.assert_array_equal(final.view(np.uint32), want.view(np.uint32))
    if name == "1138bus_sssp":
        assert w_it == int(golden("matrix")["sssp_meta"][0])  # the real reference's iteration count
    if name == "matrix5_scc":
        np.testing.assert_array_equal(want, golden("matrix5")["scc_final"])


@pytest.mark.parametrize("name,world,chunks", [("rmat11_sssp", 2, 2), ("1138bus_bfs", 3, 3), ("rmat11_bfs", 2, 4), ("matrix5_scc", 2, 5)])
def test_chunked_overlapped_all_gather_matches_single_process(name, world, chunks):
    """chunks > 1: every rank's rows in pieces, the vector laid out chunk-major, one asynchronous in-place
    all-gather per chunk issued while the next chunk is computed (the flags ride behind the last one).
    Same vectors and launch counts as the single-process loop, bit for bit."""
    case = cases()[name]
    sr, rp, ci, va, a, b = case
    x0 = O.initial_vector(sr, len(rp) - 1)
    want, w_it, w_conv = O.iterate(sr, rp, ci, va, x0, initial_y(sr, x0), a, b, 1e-4, 500)
    for rank, final, iters, conv in run_world(world, case, chunks):
        assert (iters, conv) == (w_it, w_conv), f"rank {rank}"
        np.testing.assert_array_equal(final.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("name,world,chunks", [("rmat11_sssp", 2, 1), ("1138bus_bfs", 3, 3), ("matrix5_scc", 2, 2)])
def test_direct_fan_out_exchange_matches_single_process(name, world, chunks):
    """exchange="p2p" (SH_EXCHANGE=p2p): every rank sends its finished piece straight to every peer and receives theirs
    in place (one grouped batch of isend / irecv per piece) instead of one all-gather -- the fallback SURVEY.md 5 / 8e
    names should RCCL pick a ring over xGMI.  Same bytes in the same places: bit-identical vectors and launch counts."""
    case = cases()[name]
    sr, rp, ci, va, a, b = case
    x0 = O.initial_vector(sr, len(rp) - 1)
    want, w_it, w_conv = O.iterate(sr, rp, ci, va, x0, initial_y(sr, x0), a, b, 1e-4, 500)
    for rank, final, iters, conv in run_world(world, case, chunks, exchange="p2p"):
        assert (iters, conv) == (w_it, w_conv), f"rank {rank}"
        np.testing.assert_array_equal(final.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("exchange", ["collective", "p2p"])
def test_process_group_of_one_rank_still_runs_its_collectives(exchange):
    """A group of ONE rank executes the driver's exchange code (in-place all-gather per piece / an empty fan-out) instead
    of skipping it: what the GPU test of the nccl branch relies on."""
    case = cases()["rmat11_bfs"]
    sr, rp, ci, va, a, b = case
    x0 = O.initial_vector(sr, len(rp) - 1)
    want, w_it, w_conv = O.iterate(sr, rp, ci, va, x0, x0, a, b, 1e-4, 500)
    (rank, final, iters, conv), = run_world(1, case, 3, exchange=exchange)
    assert (iters, conv) == (w_it, w_conv)
    np.testing.assert_array_equal(final.view(np.uint32), want.view(np.uint32))


def test_world1_driver_equals_oracle():
    sr, rp, ci, va, a, b = cases()["rmat11_sssp"]
    n = len(rp) - 1
    plan = ShardPlan(rp, ci, va, 0, 1)
    x0 = O.initial_vector(sr, n)
    final, iters, conv = ShardedIteration(plan, sr, OracleLocalStep(plan, sr)).run(x0, x0, a, b)
    want, w_it, w_conv = O.iterate(sr, rp, ci, va, x0, x0, a, b)
    assert (iters, conv) == (w_it, w_conv)
    np.testing.assert_array_equal(final.view(np.uint32), want.view(np.uint32))


def test_row_bounds_balance_nnz():
    rp, ci, va = H.powerlaw(100_000, 2_000_000, seed=3)
    for parts in (1, 2, 4, 8):
        b = partition.row_bounds(rp, parts)
        assert b[0] == 0 and b[-1] == 100_000 and (np.diff(b) >= 0).all() and len(b) == parts + 1
        share = np.diff(rp[b]).astype(np.float64)
        longest = np.diff(rp).max()
        assert share.max() <= 2_000_000 / parts + longest  # within one row of perfect balance
    # degenerate: more parts than rows, empty matrix
    b = partition.row_bounds(np.array([0, 5], np.int32), 4)
    assert b.tolist()[0] == 0 and b.tolist()[-1] == 1
    assert partition.row_bounds(np.zeros(1, np.int32), 2).tolist() == [0, 0, 0]


@pytest.mark.parametrize("chunks", [2, 3, 5])
def test_chunked_layout_roundtrip_and_remap(chunks):
    bounds = np.array([0, 5, 5, 130, 200, 1000])
    lay = partition.SlottedLayout(bounds, chunks)
    assert sum(lay.region(c)[1] for c in range(chunks)) == lay.length and lay.piece % 64 == 0
    v = np.arange(1000, dtype=np.float32)
    s = lay.scatter(v, np.float32(-1))
    np.testing.assert_array_equal(lay.gather(s), v)
    idx = np.array([0, 4, 5, 129, 130, 199, 999, -1, 1000, 10**6])
    pos = lay.to_slotted_index(idx)
    assert pos[-3:].tolist() == [-1, -1, -1]
    np.testing.assert_array_equal(s[pos[:7]], v[idx[:7]])
    flags = [lay.flag_index(k) for k in range(5)]
    start, length = lay.region(chunks - 1)
    assert len(set(flags)) == 5 and all(start <= f < start + length for f in flags)   # the flags ride in the last region
    for k in range(5):   # a rank's pieces tile its row range
        assert sum(lay.piece_rows(k, c)[1] for c in range(chunks)) == bounds[k + 1] - bounds[k]


def test_slotted_layout_roundtrip_and_remap():
    bounds = np.array([0, 5, 5, 130, 200])
    lay = partition.SlottedLayout(bounds)
    assert lay.payload == 128 and lay.slot == 192 and lay.length == 4 * 192
    v = np.arange(200, dtype=np.float32)
    s = lay.scatter(v, np.float32(-1))
    np.testing.assert_array_equal(lay.gather(s), v)
    idx = np.array([0, 4, 5, 129, 130, 199, -1, 200, 10**6])
    pos = lay.to_slotted_index(idx)
    assert pos[-3:].tolist() == [-1, -1, -1]          # out of range stays out of range -> identity
    np.testing.assert_array_equal(s[pos[:6]], v[idx[:6]])
    assert lay.flags(s.view(np.int32)).tolist() == [0, 0, 0, 0]
    rp, ci, va = H.rmat(8, seed=1)
    r0, r1 = 40, 90
    srp, sci, sva = partition.take_rows(rp, ci, va, r0, r1)
    assert srp[0] == 0 and srp[-1] == rp[r1] - rp[r0] and len(sci) == srp[-1]
