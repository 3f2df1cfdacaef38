"""Row-sharded multi-GPU driver for the iterative apps (SSSP, BFS): one process
per GPU, the new vector exchanged by all-gather every iteration.

The reference iterates on a single device and copies the whole vector to the
host every iteration to test convergence (app/sssp.cpp:97-176).  Here each
rank owns a contiguous, work-balanced row range of the matrix -- ONE device
matrix, one execution plan -- and a full replica of x in the layout of
partition.SlottedLayout, in which the rank's rows are cut into `chunks` pieces:

    iteration k on rank r:
      1. clear my "changed" word in x_next
      2. ONE launch of the local step (sh_spmv_step_pieces): x_next[my pieces] = kernel(A_r, x_cur, y = x_cur[my pieces]);
         the HIP kernel raises my changed word (fused convergence test) and REPORTS every piece the moment its rows
         are written (a release at system scope + a word in host memory), pieces completing in ascending order
      3. for each piece c, as soon as it is reported: all_gather_into_tensor(region c of x_next, my piece c),
         asynchronously on a side stream -- RCCL over xGMI, in place -- WHILE the launch is still computing the later
         pieces, so only the last piece's all-gather (1/chunks of the bytes, carrying the changed words) is exposed
      4. wait for the all-gathers; every rank now holds every rank's changed word: stop when all are 0
      5. swap x_cur / x_next

(Round 2 gave every piece a device matrix of its own -- two launches and a full x-tile staging per piece: +21 % per
iteration at 4 pieces on one GPU.  One plan with in-launch reporting costs a store drain, a barrier and one release
per piece and workgroup.)  Step 4 reads `parts` words back to the host (the only PCIe traffic per
iteration besides the report words) and waits for exactly that copy.  Results are bit-identical to the
single-GPU sh_iterate for every world size and chunk count, because each row
is reduced by the same code over the same data.

Variants of steps 2-4 (round 4): with ONE piece per rank nothing is reported or polled -- the exchange is enqueued behind
the launch on the launch's own stream, and iteration k + 1 is enqueued before the host has read the flags of iteration
k, behind a device word holding their OR (sh_row_pieces::gate: a launch whose gate reads 0 writes nothing), so the
host's work runs under the device's; `exchange="p2p"` (SH_EXCHANGE) replaces the all-gather by a grouped batch of
isend / irecv pairs, every rank's piece straight to every peer (the direct xGMI fan-out of SURVEY.md 5 / 8e).

`LocalStep` is the seam between this driver and the device: `HipLocalStep`
(the product) calls the C ABI on torch-owned device memory; the CPU tests plug
the oracle in instead to exercise the sharding + collective logic under gloo.
"""
import os

import numpy as np

from . import partition
from .abi import MAX_MIN_I32, MIN_PLUS_F32, OR_AND_I32, PLUS_TIMES_F32  # noqa: F401


def _torch_dtype(semiring):
    import torch
    return torch.int32 if semiring in (OR_AND_I32, MAX_MIN_I32) else torch.float32


def _np_dtype(semiring):
    return np.int32 if semiring in (OR_AND_I32, MAX_MIN_I32) else np.float32


class ShardPlan:
    """What one rank needs: its rows of the matrix (columns remapped to layout positions) and how they are cut
    into `chunks` pieces of the vector layout."""

    def __init__(self, row_ptr, col_idx, val, rank, world, chunks=1, semiring=None):
        self.rank, self.world = rank, world
        self.rows_total = len(row_ptr) - 1
        # Ranges of equal estimated work under the plan a rank's shard runs on: the engine's own weights for the x-tiled /
        # CSR-stream plans; equal entries when the shards go up in the bit-blocked (or,and) layout (HipLocalStep below:
        # semiring known, >= 2^22 entries per shard) -- 4 B per entry there, no heavy / light split.
        bits_shards = semiring == OR_AND_I32 and int(row_ptr[-1]) // max(world, 1) >= 1 << 22
        self.bounds = partition.row_bounds(row_ptr, world, cols=None if bits_shards else self.rows_total)
        self.layout = partition.SlottedLayout(self.bounds, chunks)
        self.chunks = self.layout.chunks
        self.r0, self.r1 = int(self.bounds[rank]), int(self.bounds[rank + 1])
        self.rows = self.r1 - self.r0
        # the rank's shard as ONE matrix
        rp, ci, va = partition.take_rows(row_ptr, col_idx, val, self.r0, self.r1)
        self.row_ptr, self.col_idx, self.val = np.ascontiguousarray(rp), self.layout.to_slotted_index(ci), np.ascontiguousarray(va)
        self._pieces = None

    @property
    def pieces(self):
        """Per chunk: (row_ptr, col_idx in layout positions, val, rows) -- the CPU test double computes piece by piece."""
        if self._pieces is None:
            self._pieces = []
            for c in range(self.chunks):
                lo, n = self.layout.piece_rows(self.rank, c)
                rp, ci, va = partition.take_rows(self.row_ptr, self.col_idx, self.val, lo, lo + n)
                self._pieces.append((np.ascontiguousarray(rp), ci, va, n))
        return self._pieces


class HipLocalStep:
    """Local step on the GPU through the C ABI (sh_spmv_step_pieces) on torch-owned buffers: one device matrix per
    rank, one launch per iteration, pieces reported as they complete."""

    MAX_PIECES = 8

    def __init__(self, plan, semiring, device_index):
        import ctypes as C

        import torch

        from . import abi
        from .engine import Engine
        if plan.chunks > self.MAX_PIECES:
            raise ValueError(f"at most {self.MAX_PIECES} chunks per rank")
        self.torch, self.C, self.abi = torch, C, abi
        self.plan, self.semiring = plan, semiring
        self.engine = Engine(device_index, stream=torch.cuda.current_stream().cuda_stream)
        # (a large (or,and) shard goes up in the bit-blocked layout only, as in the BFS harness; SH_OR_AND_BITS overrides)
        import os
        bits = {"or_and_bits": 2} if (semiring == OR_AND_I32 and "SH_OR_AND_BITS" not in os.environ and len(plan.col_idx) >= 1 << 22) else {}
        self.A = self.engine.upload_csr(plan.rows, plan.layout.length, plan.row_ptr, plan.col_idx,
                                        np.ascontiguousarray(plan.val, _np_dtype(semiring)), **bits)
        self.device = torch.device("cuda", device_index)
        self._wrapped = {}   # data_ptr -> engine vector handles (the driver ping-pongs two buffers)
        lay, k = plan.layout, plan.rank
        self.pc = abi.sh_row_pieces()
        # One piece: nothing to overlap, so nothing to report -- the exchange is simply enqueued behind the launch on the
        # same stream and the host never polls (its per-iteration work then runs under the launch instead of behind it:
        # 0.52 -> 0.3x ms per SSSP iteration on R-MAT-23 with a one-rank RCCL group).  Several pieces: the launch reports
        # each one and the host starts its exchange on a side stream while the later pieces are still being computed.
        self.reporting = plan.chunks > 1
        self.pc.n_pieces, self.pc.piece_rows, self.pc.report = plan.chunks, max(lay.piece, 1), 1 if self.reporting else 0
        for c in range(plan.chunks):
            self.pc.element_of_piece[c] = lay.piece_offset(k, c)
        self.round, self.words = 0, None

    def _vec(self, ptr, n):
        v = self._wrapped.get((ptr, n))
        if v is None:
            v = self._wrapped[(ptr, n)] = self.engine.wrap(ptr, n)
        return v

    def launch(self, x_cur, y_vec, x_next, alpha, beta, delta, gate=None):
        """All pieces of this rank in one launch: x_next[pieces] = kernel(A, x_cur, y_vec[pieces]); raises the rank's
        changed word; y_vec is a vector in the layout of x (x_cur itself from the second iteration on).  gate: a device
        int32 tensor or None -- a launch whose gate word is 0 when it starts writes nothing (sh_row_pieces::gate)."""
        C, lay, k = self.C, self.plan.layout, self.plan.rank
        self.pc.gate = gate.data_ptr() if gate is not None else None
        flag = self._vec(x_next.data_ptr() + lay.flag_index(k) * 4, lay.FLAG_PAD)
        flag.fill(0, np.int32)   # clear my changed word (async, same stream)
        if self.plan.rows == 0:
            self.round = None
            return
        x = self._vec(x_cur.data_ptr(), lay.length)
        y = self._vec(y_vec.data_ptr(), lay.length)
        out = self._vec(x_next.data_ptr(), lay.length)
        dt = _np_dtype(self.semiring)
        a, b = np.array([alpha], dt), np.array([beta], dt)
        rnd, words = C.c_uint32(), C.POINTER(C.c_uint32)()
        self.engine._chk(self.abi.load().sh_spmv_step_pieces(
            self.engine.h, self.semiring, self.A.h, x.h, y.h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
            out.h, C.byref(self.pc), delta, C.c_void_p(flag.device_ptr), C.byref(rnd), C.byref(words)))
        self.round = rnd.value if self.reporting else None
        self.words = np.ctypeslib.as_array(words, (self.MAX_PIECES,)) if self.reporting else None

    def piece_state(self):
        """What the engine knows about the reports of the latest launch (sh_csr_piece_state): for the error message of a
        wait that timed out."""
        C = self.C
        arr, words = (C.c_uint32 * self.MAX_PIECES)(), (C.c_uint32 * self.MAX_PIECES)()
        exp, rnd = C.c_uint32(), C.c_uint32()
        rc = self.abi.load().sh_csr_piece_state(self.engine.h, self.A.h, arr, words, C.byref(exp), C.byref(rnd))
        n = self.plan.chunks
        return {"rc": rc, "round": rnd.value, "expected_arrivals": exp.value, "host_words": list(words)[:n],
                "device_arrivals": list(arr)[:n], "layout": self.A.describe()}

    def wait_piece(self, c, timeout_s=30.0):
        """Host-side wait until piece c of the launch in flight is written and visible system-wide.  The word carries the
        round number of the launch that completed the piece (rounds only grow)."""
        if self.round is None:
            return
        import time
        t0 = time.perf_counter()
        while self.words[c] < self.round:   # (a word in host memory the reporting launch writes)
            if time.perf_counter() - t0 > timeout_s:
                raise TimeoutError(f"rank {self.plan.rank}: piece {c} of round {self.round} was not reported within "
                                   f"{timeout_s} s: {self.piece_state()}")


class ShardedIteration:
    """The do/while of the iterative apps over `world` ranks (torch.distributed must be initialised
    when world > 1).  Tensors live wherever `local.device` says (cuda for HIP, cpu for the tests)."""

    def __init__(self, plan, semiring, local, exchange=None):
        """exchange: how a finished piece reaches the other ranks --
        "collective" (default): one in-place all_gather_into_tensor per piece (RCCL picks the algorithm);
        "p2p": the direct fan-out SURVEY.md 5 / 8e names as the fallback should RCCL pick a ring (7 steps of 1/8 of the
               bytes over ONE link each, ~190 us for 33.5 MB, against ~27 us when all 7 xGMI links of a GPU carry its
               piece at once): one grouped batch of isend / irecv pairs, every rank sending its piece straight to every
               peer and receiving theirs in place -- point-to-point transfers RCCL runs concurrently, one per link.
        SH_EXCHANGE=collective|p2p overrides.  Both are bit-identical by construction (the same bytes land in the same
        places); which is faster on eight GPUs is the driver's measurement to make."""
        import os
        self.plan, self.semiring, self.local = plan, semiring, local
        self.exchange = os.environ.get("SH_EXCHANGE") or exchange or "collective"
        if self.exchange not in ("collective", "p2p"):
            raise ValueError(f"exchange must be 'collective' or 'p2p', not {self.exchange!r}")

    def run(self, x0, y0, alpha, beta, delta=1e-4, max_iters=10000):
        import time

        import torch
        import torch.distributed as dist
        plan, lay = self.plan, self.plan.layout
        k, world, chunks = plan.rank, plan.world, plan.chunks
        dt = _np_dtype(self.semiring)
        dev = self.local.device
        x0 = np.ascontiguousarray(x0, dt)
        y0 = np.ascontiguousarray(y0, dt)
        fill = x0.dtype.type(0)
        x_cur = torch.from_numpy(lay.scatter(x0, fill)).to(dev)
        x_next = torch.zeros_like(x_cur)
        y_lay = torch.from_numpy(lay.scatter(y0, fill)).to(dev)   # y of the first iteration, in the layout of x
        # (a local step that does not report its pieces -- one piece per rank -- is followed on its own stream)
        in_order = dev.type == "cuda" and not getattr(self.local, "reporting", True)
        side = (torch.cuda.current_stream(dev) if in_order else torch.cuda.Stream(device=dev)) if dev.type == "cuda" else None
        flag_i = lay.flag_index(k)
        flag_idx = torch.tensor([lay.flag_index(j) for j in range(world)], device=dev, dtype=torch.long)
        flags_dev = torch.zeros(world, dtype=torch.int32, device=dev)
        flags_host = torch.zeros(world, dtype=torch.int32)
        copied = None
        if dev.type == "cuda":
            flags_host = flags_host.pin_memory()
            copied = torch.cuda.Event()
        clears_own_flag = isinstance(self.local, HipLocalStep)
        # (a process group of ONE rank still runs its collectives: the driver's code path -- side stream, in-place
        # all-gather beside the running launch -- is then exactly what N ranks execute, with nobody to talk to)
        exchanging = world > 1 or (dist.is_available() and dist.is_initialized())
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
        iters, converged = 0, False
        if dev.type == "cuda":
            torch.cuda.synchronize()
        t_loop = time.perf_counter()

        def fan_out(buf, my_piece, plen):
            """Direct exchange: my piece to every peer, every peer's piece into its place in `buf`."""
            ops = []
            for j in range(world):
                if j != k:
                    ops.append(dist.P2POp(dist.isend, my_piece, j))
                    ops.append(dist.P2POp(dist.irecv, buf[j * plen:(j + 1) * plen], j))
            return dist.batch_isend_irecv(ops) if ops else []

        def enqueue(x_cur, x_next, first, gate=None):
            """One iteration's work: the launch, the exchange of every piece, the wait for the exchanges."""
            if not clears_own_flag:
                x_next[flag_i:flag_i + lay.FLAG_PAD] = 0
            pending = []
            y_vec = y_lay if first else x_cur
            if gate is not None:
                self.local.launch(x_cur, y_vec, x_next, alpha, beta, delta, gate=gate)
            else:
                self.local.launch(x_cur, y_vec, x_next, alpha, beta, delta)
            for c in range(chunks):
                self.local.wait_piece(c)
                if exchanging:
                    start, length = lay.region(c)
                    plen = lay.piece_len(c)
                    region = x_next[start:start + length]
                    mine = region[k * plen:(k + 1) * plen]
                    if dev.type == "cpu":
                        mine = mine.clone()   # gloo does not take an input aliasing the output
                        if self.exchange == "p2p":
                            pending.extend(fan_out(region, mine, plen))
                        else:
                            pending.append(dist.all_gather_into_tensor(region, mine, async_op=True))
                    elif dist.get_backend() == "gloo":
                        # rehearsal on one GPU (ranks share the card, gloo has no device all-gather): through the host.
                        # The piece is complete and visible: the side stream copies it out while the launch goes on.
                        with torch.cuda.stream(side):
                            mine_h = mine.to("cpu", non_blocking=True)
                            side.synchronize()
                            region_h = torch.empty(length, dtype=region.dtype)
                            if self.exchange == "p2p":
                                region_h[k * plen:(k + 1) * plen] = mine_h
                                for w in fan_out(region_h, mine_h, plen):
                                    w.wait()
                            else:
                                dist.all_gather_into_tensor(region_h, mine_h)
                            region.copy_(region_h, non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record(side)
                            pending.append(ev)
                    else:
                        # piece c is complete and visible (the host has seen its report, or the exchange follows the launch
                        # on its own stream): it need not wait for the later pieces of a reporting launch
                        with torch.cuda.stream(side):
                            if self.exchange == "p2p":
                                pending.extend(fan_out(region, mine, plen))
                            else:
                                pending.append(dist.all_gather_into_tensor(region, mine, async_op=True))
            for w in pending:
                w.wait()   # (the main stream waits: collective work or the event behind a staged copy)

        # Run-ahead (one piece per rank on the GPU under RCCL: everything of an iteration is stream-ordered): iteration
        # i + 1 is enqueued BEFORE the host has read the flags of iteration i, behind a device word that holds their OR
        # -- its launch returns at once when that word is 0 (sh_row_pieces::gate), so a converged loop costs one empty
        # launch, and the host's enqueue work and wake-up run under the device's work instead of between iterations.
        run_ahead = in_order and exchanging and dist.get_backend() != "gloo" and os.environ.get("SH_RUN_AHEAD", "1") != "0"
        if run_ahead:
            sets = [(torch.zeros(world, dtype=torch.int32, device=dev), torch.zeros(world, dtype=torch.int32).pin_memory(),
                     torch.zeros((), dtype=torch.int32, device=dev), torch.cuda.Event()) for _ in range(2)]

            def finish(x_next, st):
                fd, fh, gate_w, ev = st
                torch.index_select(x_next.view(torch.int32), 0, flag_idx, out=fd)
                torch.amax(fd, 0, out=gate_w)
                fh.copy_(fd, non_blocking=True)
                ev.record()
            enqueue(x_cur, x_next, True)
            finish(x_next, sets[0])
            enq = 1
            while True:
                st = sets[iters & 1]
                if enq < max_iters:   # iteration `enq` reads what iteration `iters` wrote; gated by the flags of `iters`
                    enqueue(x_next, x_cur, False, gate=st[2])
                    finish(x_cur, sets[enq & 1])
                    enq += 1
                st[3].synchronize()
                iters += 1
                x_cur, x_next = x_next, x_cur
                if not bool(st[1].any()):
                    converged = True
                    break
                if iters >= max_iters:
                    break
        while not run_ahead and iters < max_iters:
            enqueue(x_cur, x_next, iters == 0)
            torch.index_select(x_next.view(torch.int32), 0, flag_idx, out=flags_dev)
            flags_host.copy_(flags_dev, non_blocking=True)
            if copied is not None:
                copied.record()
                copied.synchronize()   # exactly the flag words, not everything else on the device
            iters += 1
            x_cur, x_next = x_next, x_cur
            if not bool(flags_host.any()):
                converged = True
                break
        self.last_loop_seconds = time.perf_counter() - t_loop   # iterations only (setup/readback excluded)
        final = lay.gather(x_cur.cpu().numpy())
        return final, iters, converged
