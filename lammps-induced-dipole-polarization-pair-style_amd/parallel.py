"""Multi-GPU driver: one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI),
rows of the dipole system sharded across ranks.

The reference cannot run on more than one process (README.md:5; its pack_comm/unpack_comm are
never called).  With the dd_cutoff extension the path shards naturally:

  * every rank holds all atoms (x, q, alpha: 64 B/atom, 33 MB at 512k atoms) and OWNS a contiguous
    row range [lo, hi) (spatial slabs when the atoms are ordered slab by slab);
  * list build, LJ+coul (full list rows), static field, sweeps and forces run for owned rows only;
  * the one real exchange is the dipoles: after the initial guess and after every sweep each rank
    publishes the mu (24 B/atom) of its HALO atoms -- the owned atoms that lie within the cutoff of
    another rank's extent (``HaloPlan``; for z slabs ~1/3 of the rows) -- with one all-gather of
    the packed halos; ranks whose rows are not slab-like degrade gracefully to "halo = all rows".
    Jacobi stays *identical* to the single-GPU iteration, Gauss-Seidel becomes colour-phase GS inside
    a rank and block-Jacobi across ranks (same fixed point);
  * in precision mode one double (sum dmu^2) is all-reduced per sweep and fed back to the
    device-resident loop control; energies/virial are all-reduced once per step.

``run_step`` is written against a small backend protocol so the exchange logic is covered by
world_size-2 gloo tests on CPU (tests/test_parallel_gloo.py) with a numpy stand-in backend; the
product backend is ``HipShardBackend`` (HIP kernels through the C-ABI, no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np


# E_pol per MOF5+H2 cell of the replicated boxes (dd_cutoff = cut_coul = 12.8345, precision 1e-11) on ONE GPU: -706.9202902735 / 100 cells
# (configs[2]), -1526.94782699085 / 216 (configs[3]), -2771.127537872 / 392 (configs[4])
E_POL_PER_CELL = -7.069202902735

def split_rows(n, world):
    """Contiguous row ranges, as equal as possible."""
    base, rem = divmod(n, world)
    counts = [base + (1 if r < rem else 0) for r in range(world)]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(int)
    return counts, offs


class HipShardBackend:
    """Product backend: the HIP library driven step-wise on torch's current stream."""

    def __init__(self, pair, lo, hi, device, global_count=0):
        """``global_count``: the atom count of the WHOLE system when the handle holds only a part of it
        (compact shards) -- it is the N of the stop rule sum(dmu^2)/(3N) and must agree on all ranks."""
        import torch

        self.torch = torch
        self.pair, self.lo, self.hi = pair, lo, hi
        self.dev = torch.device("cuda", device)
        L, h = pair.L, pair.h
        pair._ck(L.polar_set_row_range(h, lo, hi))
        pair._ck(L.polar_set_global_count(h, int(global_count)))
        pair._ck(L.polar_set_list_style(h, 1))
        pair._ck(L.polar_set_stream(h, C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))
        self.n = pair.nlocal
        self.own = torch.zeros((hi - lo) * 3, dtype=torch.float64, device=self.dev)
        self.chg = torch.zeros(1, dtype=torch.float64, device=self.dev)
        st = pair.get_settings()
        self.fixed, self.max_it = bool(st.fixed_iteration), st.iterations_max
        self.zodid = bool(st.zodid)
        self.lazy_end = bool(st.fixed_iteration) and bool(st.polar_gs or st.polar_gs_ranked)

    def begin(self, eflag, vflag):
        self.pair._ck(self.pair.L.polar_step_begin(self.pair.h, eflag, vflag))

    def sweep(self):
        self.pair._ck(self.pair.L.polar_step_sweep(self.pair.h))

    def sweep_part(self, part, nparts):
        self.pair._ck(self.pair.L.polar_step_sweep_part(self.pair.h, part, nparts))

    def local_change(self):
        self.pair._ck(self.pair.L.polar_change_export(self.pair.h, C.c_void_p(self.chg.data_ptr())))
        return self.chg

    def sweep_end(self, global_change, count=1):
        ptr = C.c_void_p(global_change.data_ptr()) if global_change is not None else None
        self.pair._ck(self.pair.L.polar_step_sweep_end_n(self.pair.h, ptr, count))

    def own_mu(self):
        self.pair._ck(self.pair.L.polar_mu_gather(self.pair.h, self.lo, self.hi, C.c_void_p(self.own.data_ptr())))
        return self.own

    def gather_idx(self, idx, out):
        """out[k] = mu[idx[k]] (idx: int32 device tensor of orig ids, negative = padding)"""
        self.pair._ck(self.pair.L.polar_mu_gather_idx(self.pair.h, C.c_void_p(idx.data_ptr()), idx.numel(),
                                                      C.c_void_p(out.data_ptr())))

    def scatter_idx(self, idx, src):
        self.pair._ck(self.pair.L.polar_mu_scatter_idx(self.pair.h, C.c_void_p(idx.data_ptr()), idx.numel(),
                                                       C.c_void_p(src.data_ptr())))

    def index_tensor(self, a):
        return self.torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=self.dev)

    def set_mu(self, lo, hi, buf):
        self.pair._ck(self.pair.L.polar_mu_scatter(self.pair.h, lo, hi, C.c_void_p(buf.data_ptr())))

    def state(self):
        d, it, st = C.c_int(), C.c_int(), C.c_int()
        self.pair._ck(self.pair.L.polar_step_state(self.pair.h, C.byref(d), C.byref(it), C.byref(st)))
        return d.value, it.value, st.value

    def finish(self):
        pkg = importlib.import_module(__package__)
        res = pkg.Result()
        rc = self.pair._ck(self.pair.L.polar_step_finish(self.pair.h, C.byref(res)))
        out = pkg._result_dict(res)
        out["status"] = rc
        return out

    def new_buffer(self, n):
        return self.torch.zeros(n, dtype=self.torch.float64, device=self.dev)

    def scalars_tensor(self, vals):
        return self.torch.tensor(vals, dtype=self.torch.float64, device=self.dev)


class HaloPlan:
    """Which owned atoms other ranks can see, decided from positions alone.

    Rank q must publish atom a when a lies within ``reach`` of the axis-aligned extent of ANY other
    rank's atoms (minimum image) -- a necessary condition for a to be in one of that rank's neighbor
    lists.  Every rank evaluates the same rule on the same replicated coordinates, so the plan needs no
    communication.  ``idx_all`` is the concatenation over ranks of their padded halo lists (padding -1)."""

    def __init__(self, x, prd, offs, reach):
        world = len(offs) - 1
        x = np.asarray(x, dtype=np.float64)
        lo = np.array([x[offs[r]:offs[r + 1]].min(axis=0) for r in range(world)])
        hi = np.array([x[offs[r]:offs[r + 1]].max(axis=0) for r in range(world)])
        halos = []
        for q in range(world):
            xq = x[offs[q]:offs[q + 1]]
            need = np.zeros(len(xq), dtype=bool)
            for r in range(world):
                if r == q:
                    continue
                d2 = np.zeros(len(xq))
                for k in range(3):
                    # distance from the point to the interval [lo,hi] along k, under periodic images
                    c = 0.5 * (lo[r, k] + hi[r, k])
                    h = 0.5 * (hi[r, k] - lo[r, k])
                    d = xq[:, k] - c
                    d -= prd[k] * np.round(d / prd[k])
                    d2 += np.maximum(np.abs(d) - h, 0.0) ** 2
                need |= d2 < reach * reach
            halos.append(np.nonzero(need)[0].astype(np.int64) + offs[q])
        self.counts = [len(hh) for hh in halos]
        self.maxc = max(max(self.counts), 1)
        self.idx_all = np.full(world * self.maxc, -1, dtype=np.int32)
        for q, hh in enumerate(halos):
            self.idx_all[q * self.maxc: q * self.maxc + len(hh)] = hh
        self.world = world


class P2PHaloPlan:
    """The same visibility rule, kept PER PAIR of ranks: ``send[q][r]`` = the atoms of rank q that lie
    within ``reach`` of rank r's extent.  With z slabs only the two neighbouring slabs are peers, so
    a rank exchanges two face layers point-to-point (xGMI is point-to-point: 2 x ~145 KB per sweep at
    36k atoms per GPU) instead of taking part in an all-gather of every rank's whole halo."""

    def __init__(self, x, prd, offs, reach):
        world = len(offs) - 1
        x = np.asarray(x, dtype=np.float64)
        lo = np.array([x[offs[r]:offs[r + 1]].min(axis=0) for r in range(world)])
        hi = np.array([x[offs[r]:offs[r + 1]].max(axis=0) for r in range(world)])
        self.world = world
        self.send = [[np.zeros(0, dtype=np.int32) for _ in range(world)] for _ in range(world)]
        for q in range(world):
            xq = x[offs[q]:offs[q + 1]]
            for r in range(world):
                if r == q or len(xq) == 0 or offs[r + 1] == offs[r]:
                    continue
                d2 = np.zeros(len(xq))
                for k in range(3):
                    c = 0.5 * (lo[r, k] + hi[r, k])
                    h = 0.5 * (hi[r, k] - lo[r, k])
                    d = xq[:, k] - c
                    d -= prd[k] * np.round(d / prd[k])
                    d2 += np.maximum(np.abs(d) - h, 0.0) ** 2
                self.send[q][r] = (np.nonzero(d2 < reach * reach)[0] + offs[q]).astype(np.int32)
        self.counts = [int(sum(len(a) for a in self.send[q])) for q in range(world)]

    def peers(self, rank):
        return [r for r in range(self.world) if r != rank and (len(self.send[rank][r]) or len(self.send[r][rank]))]

    def halo_of(self, rank):
        """Global ids of the atoms rank ``rank`` receives, in the order the exchange delivers them."""
        peers = self.peers(rank)
        return np.concatenate([self.send[r][rank] for r in peers]) if peers else np.zeros(0, dtype=np.int32)


def rank_classes(peer_lists):
    """Turns of a colouring shared by the ranks (polar_dist_set_schedule): greedy colouring of the peer graph in rank order,
    so that no two peers share a class.  ``peer_lists[r]`` = the peers of rank r.  Returns (class of every rank, classes).
    A ring of an even number of slabs gets 2 classes, three mutually adjacent ranks 3."""
    cls = []
    for r, peers in enumerate(peer_lists):
        used = {cls[q] for q in peers if q < r}
        c = 0
        while c in used:
            c += 1
        cls.append(c)
    return cls, (max(cls) + 1 if cls else 0)


def ghost_map(shard):
    """(owner, shift) of the periodic images a compact shard holds behind its local atoms (polar_dist_set_ghosts): the local
    index of the atom a ghost is an image of, and the whole box vectors between them."""
    n, ng = shard.nlocal, shard.nghost
    gid = np.asarray(shard.owner)
    order = np.argsort(gid[:n], kind="stable")
    pos = np.searchsorted(gid[:n][order], gid[n:])
    pos = np.minimum(pos, max(n - 1, 0))
    owner = order[pos] if n else np.zeros(0, dtype=np.int64)
    if ng and not np.array_equal(gid[:n][owner], gid[n:]):
        raise ValueError("a ghost image whose owner the shard does not hold")
    shift = shard.x[n:] - shard.x[owner] if ng else np.zeros((0, 3))
    return owner.astype(np.int32), np.ascontiguousarray(shift)


def p2p_buffers(backend, plan, rank, compact_lo=None):
    """Device index lists and packed buffers of one rank: segments ordered by peer.
    ``compact_lo``: the handle holds only [own | halo | ghosts] (workload.compact_shard) -- own atom g
    is row g - compact_lo and the received atoms are rows n_own, n_own + 1, ... in delivery order."""
    peers = plan.peers(rank)
    out_idx = np.concatenate([plan.send[rank][r] for r in peers]) if peers else np.zeros(0, dtype=np.int32)
    in_idx = np.concatenate([plan.send[r][rank] for r in peers]) if peers else np.zeros(0, dtype=np.int32)
    if compact_lo is not None:
        n_own = backend.hi - backend.lo
        out_idx = (out_idx - compact_lo).astype(np.int32)
        in_idx = (n_own + np.arange(len(in_idx))).astype(np.int32)
    seg_out = np.concatenate([[0], np.cumsum([len(plan.send[rank][r]) for r in peers])]).astype(int)
    seg_in = np.concatenate([[0], np.cumsum([len(plan.send[r][rank]) for r in peers])]).astype(int)
    return dict(peers=peers, seg_out=seg_out, seg_in=seg_in,
                idx_out=backend.index_tensor(out_idx if len(out_idx) else np.full(1, -1, dtype=np.int32)),
                idx_in=backend.index_tensor(in_idx if len(in_idx) else np.full(1, -1, dtype=np.int32)),
                send=backend.new_buffer(max(len(out_idx), 1) * 3), recv=backend.new_buffer(max(len(in_idx), 1) * 3))


def exchange_halo_p2p(backend, dist, plan, rank, bufs):
    """Send each peer the face layer it can see and take in its layer: one grouped batch of
    isend/irecv (ncclGroupStart/End under RCCL), one pack kernel before, one unpack kernel after."""
    peers = bufs["peers"]
    if not peers:
        return
    backend.gather_idx(bufs["idx_out"], bufs["send"])
    ops = []
    for k, r in enumerate(peers):
        a, b = 3 * bufs["seg_in"][k], 3 * bufs["seg_in"][k + 1]
        if b > a:
            ops.append(dist.P2POp(dist.irecv, bufs["recv"][a:b], r))
    for k, r in enumerate(peers):
        a, b = 3 * bufs["seg_out"][k], 3 * bufs["seg_out"][k + 1]
        if b > a:
            ops.append(dist.P2POp(dist.isend, bufs["send"][a:b], r))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    backend.scatter_idx(bufs["idx_in"], bufs["recv"])


def exchange_halo(backend, dist, plan, rank, bufs):
    """Publish this rank's halo dipoles and take in everybody else's (one all-gather per call)."""
    if plan.world == 1:
        return
    m = plan.maxc
    backend.gather_idx(bufs["idx_own"], bufs["send"])
    try:
        dist.all_gather_into_tensor(bufs["recv"], bufs["send"])
    except (RuntimeError, NotImplementedError):
        dist.all_gather(list(bufs["recv"].chunk(plan.world)), bufs["send"])
    backend.scatter_idx(bufs["idx_all"], bufs["recv"])  # skips padding and own rows


def halo_buffers(backend, plan, rank):
    m = plan.maxc
    return dict(idx_own=backend.index_tensor(plan.idx_all[rank * m:(rank + 1) * m]),
                idx_all=backend.index_tensor(plan.idx_all),
                send=backend.new_buffer(m * 3), recv=backend.new_buffer(plan.world * m * 3))


def exchange_mu(backend, dist, counts, offs, rank, gather_buf):
    """All-gather the owned dipoles and scatter the other ranks' rows into the local records."""
    world = len(counts)
    if world == 1:
        return
    maxc = max(counts)
    own = backend.own_mu()
    send = gather_buf["send"]
    send[: own.numel()] = own
    try:
        dist.all_gather_into_tensor(gather_buf["recv"], send)
    except (RuntimeError, NotImplementedError):  # backends without the flat form
        dist.all_gather(list(gather_buf["recv"].chunk(world)), send)
    for r in range(world):
        if r == rank or counts[r] == 0:
            continue
        seg = gather_buf["recv"][r * maxc * 3: r * maxc * 3 + counts[r] * 3]
        backend.set_mu(int(offs[r]), int(offs[r + 1]), seg)


class SweepTimer:
    """HIP events (torch's current stream = the library's stream) around every sweep of a step, read
    after the timed region: the per-launch time of the dominant kernel for bench.py's roofline."""

    def __init__(self, torch):
        self.torch, self.pairs = torch, []

    def start(self):
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()
        self._a = e

    def stop(self):
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()
        self.pairs.append((self._a, e))

    def total_ms(self):
        return float(sum(a.elapsed_time(b) for a, b in self.pairs))


POLAR_RETRY_STEP = 2  # include/polar_mi355x.h
# sweeps per all-reduce of the stop rule.  1 = the reference's rule after every sweep (PS.cpp:1194-1210): the iteration and
# sweep counts then equal the single-process ones.  bench_distributed raises it to 2 (and says so in its JSON line).
REDUCE_EVERY = max(1, int(os.environ.get("POLAR_DIST_REDUCE_EVERY", "1")))


def run_step(backend, dist, rank, world, counts, offs, eflag=1, vflag=2, check_every=4, gather_buf=None,
             halo=None, timer=None):
    """One Pair::compute across ``world`` ranks.  Returns the globally reduced result dict.
    A rank whose neighbor rows outgrew their pitch reports POLAR_RETRY_STEP; the flag is max-reduced so
    that ALL ranks repeat the step together (collectives stay matched)."""
    for attempt in range(5):
        out = _run_step_once(backend, dist, rank, world, counts, offs, eflag, vflag, check_every, gather_buf, halo,
                             timer)
        retry = 1.0 if out.get("status") == POLAR_RETRY_STEP else 0.0
        if world > 1:
            t = backend.scalars_tensor([retry])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            retry = float(t.cpu().numpy()[0])
        if retry == 0.0:
            return out
        if timer is not None:
            timer.pairs.clear()
    raise RuntimeError("neighbor list pitch overflow persists")


class SweepGraph:
    """The per-sweep sequence of a step -- sweep launches, the all-reduced stop rule, pack / point-to-point exchange /
    unpack of the halo dipoles -- recorded ONCE as a HIP graph of `block` sweeps and replayed, so that a sweep costs
    the host one graph launch per `block` sweeps instead of ~10 Python calls (each sweep is only ~0.1-0.3 ms of GPU
    work at 65k atoms per GPU).  Everything in the body is stream-ordered work on buffers that exist before the
    capture: library kernels on torch's current stream (polar_set_stream), in-place collectives of torch.distributed
    (RCCL records its kernels into the graph).  Opt-in (POLAR_DIST_GRAPH=1): exercised here with one rank only."""

    def __init__(self, torch, backend, body, block):
        self.torch, self.block = torch, block
        self.graph = torch.cuda.CUDAGraph()
        stream = torch.cuda.current_stream(backend.dev)
        # one eager pass first: lazy initialisations (RCCL channels, kernel modules) must not happen under capture
        body()
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph, stream=stream):
            for _ in range(block):
                body()

    def replay(self):
        self.graph.replay()


def _run_step_once(backend, dist, rank, world, counts, offs, eflag, vflag, check_every, gather_buf, halo, timer):
    """``halo`` = (HaloPlan | P2PHaloPlan, buffers) switches the dipole exchange from "all owned rows" to halo rows."""
    maxc = max(counts)
    if halo is not None:
        plan, hb = halo
        p2p = isinstance(plan, P2PHaloPlan)

        def exchange_mu(backend, dist, counts, offs, rank, gather_buf):  # noqa: F811 (local override)
            if p2p:
                exchange_halo_p2p(backend, dist, plan, rank, hb)
            else:
                exchange_halo(backend, dist, plan, rank, hb)
    else:
        exchange_mu = globals()["exchange_mu"]
        if gather_buf is None:
            gather_buf = dict(send=backend.new_buffer(maxc * 3), recv=backend.new_buffer(world * maxc * 3))
    backend.begin(eflag, vflag)
    exchange_mu(backend, dist, counts, offs, rank, gather_buf)   # initial guess of the other ranks
    sweeps = 0
    keep_going = None
    use_graph = (os.environ.get("POLAR_DIST_GRAPH") == "1" and hasattr(backend, "torch") and not backend.zodid
                 and not getattr(backend, "lazy_end", False) and timer is None)
    if use_graph:
        def body():  # one sweep of the precision-mode loop below, with nothing the host has to look at
            backend.sweep()
            if world > 1:
                chg = backend.local_change()
                dist.all_reduce(chg)
                backend.sweep_end(chg)
            else:
                backend.sweep_end(None)
            exchange_mu(backend, dist, counts, offs, rank, gather_buf)

        sg = getattr(backend, "_sweep_graph", None)
        if sg is None:
            # the recording pass runs real sweeps: do it on a scratch step, then start this step again
            sg = backend._sweep_graph = SweepGraph(backend.torch, backend, body, check_every)
            backend.finish()
            backend.begin(eflag, vflag)
            exchange_mu(backend, dist, counts, offs, rank, gather_buf)
        while sweeps < backend.max_it + 1:
            sg.replay()
            sweeps += check_every
            done, _, _ = backend.state()   # sweeps past the end of a finished solve are no-ops on the device
            if done:
                break
    elif not backend.zodid:
        for sw in range(backend.max_it + 1):
            if timer is not None:
                timer.start()
            backend.sweep()
            if timer is not None:
                timer.stop()
            if world > 1 and not backend.fixed:
                # the stop rule (PS.cpp:1194-1210) needs the sum over all ranks: one all-reduced double.  It is taken on every
                # second sweep only; in between the end-of-sweep logic is told "not converged yet" (+inf).  That saves half of
                # the all-reduce latencies -- a sizeable share of a ~0.1 ms sweep -- for at most one sweep past the point where
                # the rule would have stopped (it only converges the dipoles further).  With S = sweeps x all-reduce latency
                # and C = the time of a sweep the best cadence is sqrt(2 S / C): 2.7 - 3.6 for 17 - 30 us against 170 us
                if (sw % REDUCE_EVERY) == REDUCE_EVERY - 1 or sw >= backend.max_it:
                    chg = backend.local_change()
                    dist.all_reduce(chg)
                    backend.sweep_end(chg)
                else:
                    if keep_going is None:
                        keep_going = backend.scalars_tensor([float("inf")])
                    backend.sweep_end(keep_going)
            elif getattr(backend, "lazy_end", False):
                # fixed-iteration Gauss-Seidel: no decision between sweeps -> the end-of-sweep logic of all
                # sweeps but the last in one launch, then the last (its sum |dmu|^2 is the one reported)
                last = backend.max_it
                if sw == last - 1:
                    backend.sweep_end(None, count=last)
                elif sw == last or last == 0:
                    backend.sweep_end(None)
            else:
                backend.sweep_end(None)
            exchange_mu(backend, dist, counts, offs, rank, gather_buf)
            sweeps += 1
            if not backend.fixed and (sw % check_every) == check_every - 1:
                done, _, _ = backend.state()
                if done:
                    break
    out = backend.finish()
    if world > 1:
        keys = ["eng_vdwl", "eng_coul", "eng_pol", "u_self", "u_ef", "u_dd"]
        vals = [out[k] for k in keys] + list(out["virial"]) + [float(out["dd_pairs"])]
        t = backend.scalars_tensor(vals)
        dist.all_reduce(t)
        v = t.cpu().numpy()
        for k, x in zip(keys, v[:6]):
            out[k] = float(x)
        out["virial"] = v[6:12].copy()
        out["dd_pairs"] = int(v[12])
    out["driver_sweeps"] = sweeps
    return out


class HostStagedDist:
    """torch.distributed with every device tensor staged through the host (REHEARSAL ONLY, POLAR_DIST_BACKEND=gloo):
    lets several ranks share the one GPU of a test box -- RCCL refuses two ranks on one device -- so that the whole
    of bench_distributed (shard construction, halo plans, device-built lists of the own rows, the sweep protocol,
    the retry path, the JSON line) runs with world > 1.  Only the RCCL calls themselves stay unrehearsed."""

    def __init__(self, dist):
        self._d = dist
        self.ReduceOp = dist.ReduceOp
        self.isend, self.irecv = "isend", "irecv"

    class _Op:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Req:
        def __init__(self, req, host, dev):
            self.req, self.host, self.dev = req, host, dev

        def wait(self):
            self.req.wait()
            if self.dev is not None:
                self.dev.copy_(self.host)

    def P2POp(self, op, tensor, peer):
        return HostStagedDist._Op(op, tensor, peer)

    def batch_isend_irecv(self, ops):
        reqs = []
        for o in ops:
            host = o.tensor.detach().cpu().contiguous()
            if o.op == "isend":
                reqs.append(HostStagedDist._Req(self._d.isend(host, o.peer), host, None))
            else:
                reqs.append(HostStagedDist._Req(self._d.irecv(host, o.peer), host, o.tensor))
        return reqs

    def all_reduce(self, t, op=None):
        host = t.detach().cpu()
        self._d.all_reduce(host, op=op if op is not None else self._d.ReduceOp.SUM)
        t.copy_(host)

    def all_gather_into_tensor(self, out, inp):
        ho, hi = out.detach().cpu(), inp.detach().cpu()
        self._d.all_gather(list(ho.chunk(self._d.get_world_size())), hi)
        out.copy_(ho)

    def barrier(self):
        self._d.barrier()

    def destroy_process_group(self):
        self._d.destroy_process_group()


# --------------------------------------------------------------------------------------------
# DESIGN section 6's projection of the N-rank step, by (ranks, schedule): sweeps from the emulation on mock ranks, sweep-kernel
# time per sweep from one shard stepped alone, exposed exchange + all-reduce per sweep from the one-rank RCCL runs (+ ~13 us
# of link assumed), non-sweep time of a rank's step.  The first hardware run is compared with these (config.calibration.model).
MODEL = {
    (2, "legacy"): dict(ms_per_step=12.5, sweeps=38, us_sweep_kernels_per_sweep=232, us_exchange_and_stop_rule_per_sweep=36, ms_non_sweep=2.3),
    (2, "lag1"): dict(ms_per_step=12.1, sweeps=35, us_sweep_kernels_per_sweep=232, us_exchange_and_stop_rule_per_sweep=48, ms_non_sweep=2.3),
    (4, "legacy"): dict(ms_per_step=7.8, sweeps=38, us_sweep_kernels_per_sweep=129, us_exchange_and_stop_rule_per_sweep=36, ms_non_sweep=1.5),
    (8, "legacy"): dict(ms_per_step=6.8, sweeps=38, us_sweep_kernels_per_sweep=108, us_exchange_and_stop_rule_per_sweep=34, ms_non_sweep=1.38),
    (8, "lag1"): dict(ms_per_step=7.8, sweeps=34, us_sweep_kernels_per_sweep=108, us_exchange_and_stop_rule_per_sweep=80, ms_non_sweep=1.38),
    # (23 sweeps: the driver on 8 mock ranks; 108 us of sweep kernels and 21 us of mixing per sweep: one shard stepped alone on the round-5 library -- profiles/r05_lab_shards_n8_accel.txt)
    (8, "legacy_accel4"): dict(ms_per_step=5.2, sweeps=23, us_sweep_kernels_per_sweep=108, us_exchange_and_stop_rule_per_sweep=37, us_accel_mixing_per_sweep=21, ms_non_sweep=1.38),
}


class RankJob:
    """One rank of `bench.py --gpus N`: the shard of the box this rank owns, and -- per schedule -- a handle, the library's RCCL
    driver (or, for the rehearsal backends, the Python sweep loop) and the measurements."""

    def __init__(self, args, rank, world, local_rank, torch, dist, backend_name):
        self.args, self.rank, self.world, self.local_rank = args, rank, world, local_rank
        self.torch, self.dist, self.backend_name = torch, dist, backend_name
        self.pkg = importlib.import_module(__package__)
        self.wl = importlib.import_module(__package__ + ".workload")
        import bench as B  # repo root is on sys.path (bench.py put it there)

        self.B = B
        self.p = self.be = self.driver = None
        self.driver_fallback = None

    # -- the box and this rank's share of it (once) --------------------------------------------------------------------------
    def build_shard(self, keywords):
        args, world, rank, wl, B = self.args, self.world, self.rank, self.wl, self.B
        k = args.config or (4 if world >= 8 else (3 if world > 1 else 2))
        self.cfg = cfg = dict(B.CONFIGS[k])
        if args.reps:
            cfg["reps"] = tuple(args.reps)
        self.reps = cfg["reps"]
        sg = B.build_workload(wl, self.reps, list(args.extra) + list(keywords), build_list=False, solver=cfg["solver"])   # atoms + ghosts, no host list
        self.n_total = sg.nlocal
        if os.environ.get("POLAR_DIST_SPLIT", "slabs") == "rows":
            counts, offs = split_rows(self.n_total, world)   # equal row ranges of the replica order (cuts through replica cells)
        else:
            # geometric z slabs of equal atom counts; a sorbate molecule and a cluster of bonded framework atoms (< 1.6 A
            # apart: the strongest couplings) stay on one rank (workload.slab_order): two peers per rank and the thinnest halo
            # whatever the number of ranks, and 37 instead of 40 sweeps on 8 slabs
            glue_dist = float(os.environ.get("POLAR_DIST_GLUE", "1.6"))
            grid = [int(v) for v in os.environ.get("POLAR_DIST_GRID", "").replace("x", ",").split(",") if v]
            if len(grid) == 3 and grid[0] * grid[1] * grid[2] == world and (grid[0] > 1 or grid[1] > 1):
                order, offs = wl.brick_order(sg, grid, glue_dist=glue_dist)   # bricks (opt-in): up to 7 peers per rank at 2 x 2 x 2
                sg = wl.permute_locals(sg, order)
                counts = [int(offs[r + 1] - offs[r]) for r in range(world)]
            else:
                order, key, glue = wl.slab_order(sg, axis=2, glue_dist=glue_dist)
                sg = wl.permute_locals(sg, order)
                counts, offs = wl.split_sorted(key[order], world, glue)
        self.counts, self.offs = counts, offs
        self.lo, self.hi = int(offs[rank]), int(offs[rank + 1])
        # reach = the neighbor-list cutoff of the LJ/Coulomb rows (max cut + skin) -- it covers the dd cutoff
        self.reach = float(sg.extra["cutneigh"]) + 1e-6
        self.mode = os.environ.get("POLAR_HALO_MODE", "p2p")
        if self.mode == "allgather":
            self.s = sg   # every rank holds ALL atoms and owns the rows [lo, hi); one all-gather of the halos per sweep
            self.plan = HaloPlan(sg.x[:sg.nlocal], sg.prd, offs, self.reach)
            self.rows_own = int(np.count_nonzero(sg.alpha[self.lo:self.hi]))
        else:
            # default: every rank holds only [own | halo | ghosts] (per-step cost independent of the number of ranks) and
            # exchanges face layers point-to-point with its slab neighbours
            self.plan = P2PHaloPlan(sg.x[:sg.nlocal], sg.prd, offs, self.reach)
            self.s = wl.compact_shard_geometric(sg, np.arange(self.lo, self.hi), self.plan.halo_of(rank), self.reach)
            self.rows_own = int(np.count_nonzero(self.s.alpha[:self.hi - self.lo]))
        self.n_held = self.s.nlocal + self.s.nghost

    # -- one schedule: handle + driver ----------------------------------------------------------------------------------------
    def open(self, name, lag, split_comm, accel=0):
        """(Re)create the handle and the driver for schedule ``name``.  Collective: every rank, same arguments."""
        self.close()
        torch, dist, pkg, rank, world, local_rank = self.torch, self.dist, self.pkg, self.rank, self.world, self.local_rank
        s, lo, hi, plan = self.s, self.lo, self.hi, self.plan
        if accel and s.settings.polar_accel != accel:
            import copy
            import dataclasses
            s = copy.copy(s)
            s.settings = dataclasses.replace(s.settings, polar_accel=int(accel))
        self.name, self.lag, self.ncls = name, lag, 0
        if self.mode == "allgather":
            self.p = pkg.pair_from_system(s, device=local_rank, device_neigh=True, row_range=(lo, hi))
            self.be = HipShardBackend(self.p, lo, hi, local_rank)
            self.halo = (plan, halo_buffers(self.be, plan, rank)) if world > 1 else None
        else:
            self.p = pkg.pair_from_system(s, device=local_rank, device_neigh=True, row_range=(0, hi - lo))
            self.be = HipShardBackend(self.p, 0, hi - lo, local_rank, global_count=self.n_total)
            self.halo = (plan, p2p_buffers(self.be, plan, rank, compact_lo=lo))
        be = self.be
        if world > 1:
            # establish the connections of torch's communicator (peer-to-peer channels are created lazily at first use) outside
            # any timed region: one exchange of the still empty buffers plus one tiny all-reduce
            if isinstance(plan, P2PHaloPlan):
                hb = self.halo[1]
                ops = []
                for kk, r in enumerate(hb["peers"]):
                    a, b = 3 * hb["seg_in"][kk], 3 * hb["seg_in"][kk + 1]
                    if b > a:
                        ops.append(dist.P2POp(dist.irecv, hb["recv"][a:b], r))
                for kk, r in enumerate(hb["peers"]):
                    a, b = 3 * hb["seg_out"][kk], 3 * hb["seg_out"][kk + 1]
                    if b > a:
                        ops.append(dist.P2POp(dist.isend, hb["send"][a:b], r))
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            t = be.scalars_tensor([0.0])
            dist.all_reduce(t)
            torch.cuda.synchronize()
        # The per-sweep loop: the library's own C++ driver (polar_dist_*: RCCL calls enqueued by the library, no Python between
        # two sweeps) whenever the backend is RCCL and the exchange is point-to-point; the Python loop (run_step) remains for the
        # rehearsal backends (gloo: ranks sharing a GPU, exchanges staged through the host) and for the all-gather variant.
        self.driver, self.driver_fallback = None, None
        if self.backend_name == "nccl" and isinstance(plan, P2PHaloPlan) and os.environ.get("POLAR_DIST_DRIVER", "cpp") == "cpp":
            ids = [pkg.PolarDist.unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ids, src=0)
            driver_error = None
            try:
                if split_comm:
                    os.environ["POLAR_DIST_SPLIT_COMM"] = "1"   # (read by polar_dist_create: a communicator of its own for the all-reduces)
                else:
                    os.environ.pop("POLAR_DIST_SPLIT_COMM", None)
                self.driver = pkg.PolarDist(ids[0], rank, world, device=local_rank)
                peers = plan.peers(rank)
                n_own = hi - lo
                send_lists = [(np.asarray(plan.send[rank][r]) - lo).astype(np.int32) for r in peers]
                recv_lists, at = [], n_own
                for r in peers:
                    m = len(plan.send[r][rank])
                    recv_lists.append(np.arange(at, at + m, dtype=np.int32))
                    at += m
                self.driver.set_halo(self.p, peers, send_lists, recv_lists)
                self.driver.set_cadence(REDUCE_EVERY, 4)
                cls, self.ncls = rank_classes([plan.peers(r) for r in range(world)])
                self.driver.set_schedule(lag, cls[rank], self.ncls if lag >= 0 else 0)
                self.driver.set_ghosts(self.p, *ghost_map(s))
            except Exception as e:  # noqa: BLE001 -- the communicator could not be made on this rank: all ranks must take the same loop
                driver_error = repr(e)
            ok = torch.tensor([0.0 if driver_error else 1.0], dtype=torch.float64, device=be.dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:  # same HIP kernels, the sweep loop in Python over torch.distributed instead (said in the JSON line)
                if self.driver is not None and not driver_error:
                    self.driver.close()
                self.driver = None
                self.driver_fallback = driver_error or "another rank could not create the RCCL communicator"
                print(f"[rank {rank}] polar_dist driver unavailable ({self.driver_fallback}): Python sweep loop", file=sys.stderr, flush=True)

    def close(self):
        if self.driver is not None:
            self.driver.close()
        if self.p is not None:
            self.p.close()
        self.driver = self.p = self.be = None

    def one_step(self, timer=None):
        if self.driver is not None:
            return self.driver.step(self.p, 1, 2)
        return run_step(self.be, self.dist, self.rank, self.world, self.counts, self.offs, gather_buf=None, halo=self.halo, timer=timer)

    def timed(self, steps, warmup):
        """`warmup` untimed steps (at least one: lists, colours, connections), then EXACTLY `steps` steps between barrier +
        synchronize on both sides; the MAX over the ranks of the wall time."""
        torch, dist = self.torch, self.dist
        for _ in range(max(warmup, 1)):
            out = self.one_step()
        dist.barrier()
        torch.cuda.synchronize()
        timer = SweepTimer(torch)
        t0 = time.perf_counter()
        for it in range(steps):  # the event pairs cost ~5 % of a step: only the last timed step carries them (Python loop only)
            out = self.one_step(timer if (it == steps - 1 and self.driver is None) else None)
        torch.cuda.synchronize()
        dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.be.dev)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return out, float(dt.item()), timer

    def calibrate(self):
        """ONE more step with the library's timed events between the parts of its sweep loop (outside any timed region), then the
        MAX over the ranks of every figure: what DESIGN section 6's model  t = sweeps x (sweep kernels + exposed exchange + stop
        rule) + non-sweep  is calibrated on.  None without the in-library driver."""
        if self.driver is None:
            return None
        torch, dist = self.torch, self.dist
        self.driver.profile(True)
        try:
            o = self.driver.step(self.p, 1, 2)
            pr = self.driver.profile_get()
        finally:
            self.driver.profile(False)
        keys = ["sweep_kernels", "exchange", "stop_rule", "other", "accel"]
        loc = [pr[k] for k in keys] + [o["ms_solve"], o["ms_total"] - o["ms_solve"], o["ms_total"], float(o["sweeps"]), o["ms_list"], o["ms_static"], o["ms_force"], o["ms_ljcoul"]]
        t = torch.tensor(loc, dtype=torch.float64, device=self.be.dev)
        tmin = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        v, vmin = t.cpu().numpy(), tmin.cpu().numpy()
        sw = max(v[8], 1.0)
        exposed = max(v[5] - v[0] - v[2] - v[4], 0.0)   # solve minus sweep kernels, stop rule and mixing: exchanges on the compute stream + waits for the communication stream + host looks
        cal = {"what": "one extra step with timed events between the parts of the sweep loop (polar_dist_profile); every figure is the MAX over the "
                       "ranks (min in *_min); exposed exchange = solve - sweep kernels - stop rule - mixing",
               "sweeps": int(v[8]), "ms_solve": v[5], "ms_non_sweep": v[6], "ms_device_step": v[7],
               "ms_sweep_kernels": v[0], "ms_exchange_on_compute_stream": v[1], "ms_stop_rule": v[2], "ms_waits_and_host_looks": v[3], "ms_accel_mixing": v[4],
               "ms_exposed_exchange": exposed,
               "us_sweep_kernels_per_sweep": 1e3 * v[0] / sw, "us_exposed_exchange_per_sweep": 1e3 * exposed / sw, "us_stop_rule_per_sweep": 1e3 * v[2] / sw,
               "ms_sweep_kernels_min": vmin[0], "ms_solve_min": vmin[5], "ms_non_sweep_min": vmin[6],
               "ms_list": v[9], "ms_static": v[10], "ms_force": v[11], "ms_ljcoul_side_stream": v[12],
               "profile_intervals": pr["intervals"]}
        model = MODEL.get((self.world, self.name))
        if model:
            cal["model"] = dict(model, source="DESIGN.md section 6 (projection from one-GPU measurements)")
        return cal

    def md_leg(self):
        """MD-shaped leg (driver only): every rank moves ITS OWN atoms each step (thermal-size jitter, inside the skin), uploads them
        (polar_set_positions_range: PCIe), fetches the halo atoms' positions from their owners over RCCL and rebuilds the ghost
        images (polar_dist_positions) -- north_star "ghost x ... over xGMI" --; on the last step the device list is rebuilt."""
        if self.driver is None:
            return None
        torch, dist, p, s = self.torch, self.dist, self.p, self.s
        rng = np.random.default_rng(1000 + self.rank)
        n_own = self.hi - self.lo
        x_own0 = np.ascontiguousarray(s.x[:n_own])
        disp = np.zeros_like(x_own0)
        md_steps, every = 10, 10
        t_steps, sw = [], []
        self.one_step()
        for k in range(md_steps):
            disp += rng.normal(scale=0.01, size=disp.shape)
            dist.barrier()
            t1 = time.perf_counter()
            p.set_positions_range(0, n_own, x_own0 + disp)
            self.driver.positions(p)
            if k % every == every - 1:
                p.build_neighbors_from_system(s)
            o2 = self.driver.step(p, 1, 2)
            torch.cuda.synchronize()
            tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=self.be.dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_steps.append(1e3 * float(tt.item())); sw.append(o2["sweeps"])
        return {"what": f"{md_steps} steps; per step every rank uploads its own atoms' new positions (jitter 0.01 A per step), halo positions and ghost images "
                        "through polar_dist_positions (RCCL), polar_dist_step; the device list rebuilt on the last step; wall clock, max over ranks",
                "ms_per_step_md": float(np.mean(t_steps[:-1])), "ms_reneighbor_step": t_steps[-1], "sweeps_per_step": float(np.mean(sw)),
                "atom_steps_per_s_md": self.n_total / (float(np.mean(t_steps[:-1])) * 1e-3)}

    def record(self, out, dt, timer, steps, warmup):
        """The JSON line of the schedule that was just timed (rank 0 prints it; every rank builds it: same numbers)."""
        B, pkg, world, plan, counts, reps = self.B, self.pkg, self.world, self.plan, self.counts, self.reps
        n_total, lag, ncls, driver = self.n_total, self.lag, self.ncls, self.driver
        # roofline of the dominant kernel on this rank (same accounting as the single-GPU line, bench.py)
        if driver is not None:  # the library's own events around the solve of the last step (exchanges and all-reduces included)
            launches = max(out["sweeps"], 1) * max(out["ncolors"], 1)
            ms_launch = out["ms_solve"] / launches
        else:
            launches = max(len(timer.pairs), 1) * max(out["ncolors"], 1)
            ms_launch = timer.total_ms() / launches
        pairs_rank = out["dd_pairs"] / world          # dd_pairs was all-reduced; equal shares of a uniform box
        bytes_launch = (4.0 * pairs_rank + 112.0 * self.rows_own) / max(out["ncolors"], 1)
        achieved = bytes_launch / max(ms_launch * 1e-3, 1e-12) / 1e9
        per_cell = out["eng_pol"] / float(np.prod(reps))
        backend_txt = "RCCL" if self.backend_name == "nccl" else "REHEARSAL: " + self.backend_name + ", host-staged, ranks sharing a GPU"
        if driver is not None and lag >= 0:
            sched = (f"one colouring shared by the ranks ({ncls} turns), colour c's boundary dipoles exchanged after phase c on a second stream, "
                     f"a phase waits for the exchange issued {lag + 1} phase(s) earlier")
        else:
            sched = "every rank colours for itself, one exchange of all halo dipoles per sweep on the compute stream (block-Jacobi across ranks)"
        if self.s.settings.polar_accel or "polar_accel" in self.args.extra or self.name.endswith("accel4"):
            sched += ("; polar_accel (Anderson mixing, the dot products on the stop rule's all-reduce)" if driver is not None
                      else "; polar_accel asked for but NOT applied: the Python sweep loop of the rehearsal backends has no mixing step")
        return {
            "metric": "atom-steps/sec", "value": n_total * steps / dt, "unit": "atom-steps/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": B.describe(self.cfg, n_total) + f"; {world} rank(s), {counts[0]} own atoms per GPU (z slabs), "
                                   f"colour-phase GS per rank + {'all-gather' if isinstance(plan, HaloPlan) else 'point-to-point exchange'} "
                                   f"of the halo dipoles ({backend_txt}), LJ/Coulomb lists built on the device",
                       "natoms": n_total, "sweeps": out["sweeps"], "iterations": out["iterations"], "colors": out["ncolors"],
                       "dd_pairs": out["dd_pairs"], "eng_pol": out["eng_pol"], "rms_dmu_last_sweep": out["rms_dmu"],
                       # a result check the line carries itself: every box here is whole copies of ONE cell wider than two cutoffs, so
                       # E_pol per cell is a property of the cell (tests/test_gpu_fullsize.py); the single-GPU value at precision 1e-11
                       # (configs[2], [3] and [4] on one GPU agree on it to 1e-9)
                       "eng_pol_per_cell": per_cell, "eng_pol_per_cell_one_gpu": E_POL_PER_CELL,
                       "eng_pol_rel_dev_from_one_gpu": abs(per_cell - E_POL_PER_CELL) / abs(E_POL_PER_CELL),
                       "ms_per_dipole_iteration": 1e3 * dt / steps / max(out["sweeps"], 1),
                       "ms_device_rank0": {k2: out[k2] for k2 in ("ms_total", "ms_list", "ms_ljcoul", "ms_static", "ms_solve", "ms_force")},
                       "atoms_held_rank0": self.n_held, "halo_rows_per_rank": plan.counts, "rows_per_rank": counts,
                       "peers_rank0": len(plan.peers(0)) if hasattr(plan, "peers") else None,
                       "stop_rule_allreduce_every_sweeps": REDUCE_EVERY,
                       "rccl_ranks": driver.comm_count() if driver is not None else None,
                       "schedule_name": self.name, "schedule": sched,
                       "sweep_loop": ("in-library C++ driver (polar_dist_step): RCCL send / receive groups and the all-reduced stop rule enqueued by the "
                                      "library, state read every 4 sweeps") if driver is not None
                                     else "Python loop over the stepwise C-ABI (torch.distributed collectives)" + (f" -- C++ driver unavailable: {self.driver_fallback}" if self.driver_fallback else ""),
                       "exchanges_last_step": out.get("exchanges"), "allreduces_last_step": out.get("allreduces"),
                       "kernel_version": pkg.kernel_version()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": None, "kernel": f"k_field_lp (dipole-field sweep, one launch per colour phase; rank 0; {pkg.kernel_version()})",
                         "bytes_per_launch": bytes_launch, "ms_per_launch": ms_launch},
        }


def bench_distributed(args, rank, world, local_rank):
    """bench.py --gpus N (N > 1): STRONG scaling on a fixed box -- BASELINE configs[3] (6x6x6 = 291,384 atoms) for
    N = 2, 4 and configs[4] (7x7x8 = 528,808 atoms) for N = 8, ranked GS to precision 1e-11 -- one rank per GPU, z slabs, every
    rank holding only [own | halo | ghosts], LJ/Coulomb lists of the own rows built on the device, halo dipoles exchanged
    point-to-point with the slab neighbours.  --schedule picks how (bench.SCHEDULES); the headline is "legacy".
    The record cannot be lost after the timed steps (bench.Emitter): the headline line exists before any extra leg starts."""
    import torch
    import torch.distributed as dist

    import bench as B  # repo root is on sys.path (bench.py put it there)

    global REDUCE_EVERY
    if "POLAR_DIST_REDUCE_EVERY" not in os.environ:
        REDUCE_EVERY = 2   # bench cadence: at most one sweep past the stop rule for half of the all-reduce latencies
    backend_name = os.environ.get("POLAR_DIST_BACKEND", "nccl")
    if backend_name == "nccl":
        if torch.cuda.device_count() <= local_rank:   # never an N = 1 line for a --gpus N request
            raise SystemExit(f"bench.py: rank {rank} has no GPU of its own ({torch.cuda.device_count()} visible, RCCL takes one rank per device)")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        # rehearsal on a box with fewer GPUs than ranks: ranks share devices, exchanges go through the host
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend_name)
        dist = HostStagedDist(dist)
    if os.environ.get("POLAR_DIST_GRAPH") == "1":
        # graph capture needs a non-default stream: everything of this run (library kernels, RCCL) goes onto one side stream
        torch.cuda.set_stream(torch.cuda.Stream(device=local_rank))

    def settings_of(name, explicit):
        """(lag, split communicator, polar_accel depth) of a schedule; without --schedule the environment may override (experiments)."""
        env, kw = B.SCHEDULES[name]
        lag = int(env.get("POLAR_DIST_LAG", "-1"))
        split = env.get("POLAR_DIST_SPLIT_COMM", "0") == "1"
        if not explicit:
            lag = int(os.environ.get("POLAR_DIST_LAG", lag))
            split = os.environ.get("POLAR_DIST_SPLIT_COMM", "1" if split else "0") == "1"
        accel = int(kw[kw.index("polar_accel") + 1]) if "polar_accel" in kw else 0
        return lag, split, accel

    name = args.schedule or "legacy"
    job = RankJob(args, rank, world, local_rank, torch, dist, backend_name)
    job.build_shard(B.SCHEDULES[name][1])
    lag, split, accel = settings_of(name, bool(args.schedule))
    job.open(name, lag, split, accel)
    out, dt, timer = job.timed(args.steps, args.warmup)
    line = job.record(out, dt, timer, args.steps, args.warmup)
    config = line["config"]
    # ---- the measured record exists from here on ----
    em = B.Emitter(rank == 0, float(os.environ.get("POLAR_BENCH_EXTRAS_BUDGET", "420")))
    em.headline(line)

    def leg(where, key, fn):
        em.stage = key
        t0 = time.time()
        try:
            where[key] = fn()
        except Exception as e:  # noqa: BLE001 -- (a rank-local exception leaves the other ranks in a collective: the watchdog ends the job)
            where[key] = {"error": repr(e), "after_s": time.time() - t0}

    leg(config, "calibration", job.calibrate)
    if not args.no_extras:
        leg(config, "md_leg", job.md_leg)
        own_launcher = bool(os.environ.get("POLAR_BENCH_LAUNCHER"))
        if not own_launcher and not args.schedule and job.driver is not None:
            # under a foreign launcher (the driver's own torch.distributed.run) nobody starts the other schedules as jobs of their
            # own: they run here, in this process, AFTER the headline is safe -- a hang costs what follows it, the watchdog prints
            config["schedules"] = {name: B.schedule_summary(line)}
            config["headline_schedule"] = name
            for other in B.SCHEDULES:
                if other == name:
                    continue

                def run_other(other=other):
                    l2, s2, a2 = settings_of(other, True)
                    job.open(other, l2, s2, a2)
                    o2, dt2, tm2 = job.timed(args.steps, args.warmup)
                    sub = job.record(o2, dt2, tm2, args.steps, args.warmup)
                    sub["config"]["calibration"] = job.calibrate()
                    return B.schedule_summary(sub)

                leg(config["schedules"], other, run_other)
    em.final(line)
    bad = not (config["eng_pol_rel_dev_from_one_gpu"] <= B.EPOL_TOL)
    # the line is out: nothing below may keep the process (and with it the whole job) alive -- tearing communicators down
    # can wait for a peer that an earlier leg lost
    import threading
    guard = threading.Timer(60.0, lambda: os._exit(4 if bad else 0))
    guard.daemon = True
    guard.start()
    try:
        job.close()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass
    guard.cancel()
    if bad:   # every rank holds the same all-reduced energy: the same exit code everywhere
        if rank == 0:
            print(f"bench.py: E_pol per cell is {config['eng_pol_rel_dev_from_one_gpu']:.2e} off the one-GPU value: wrong result", file=sys.stderr)
        sys.exit(4)
