"""One particle filter sharded over the GPUs of a node: one process per GPU, each engine holds a
contiguous slice of the particle index space, `torch.distributed` (backend "nccl" = RCCL over
xGMI) carries the three small exchanges the path really has:

  update_action    none: every rank walks the same drand48 stream (attempt compaction over the GLOBAL
                   Gaussian ranks), materialises the draws of its own index range and ends on the same state
  update_sensor    all-gather of W per-shard weight totals (8 B each)
  update_resample  per candidate-draw window one integer all-reduce(sum) of [6, window] int64 (pose
                   bits + histogram key of every draw; exactly one shard writes each column, the
                   others contribute 0).  The shards' slices of the global CDF come from the totals
                   gathered by the sensor update (slice_q = total_q / sum(totals), the same quotient on
                   every rank), so no further exchange is needed; only when the weights were set by
                   something else than update_sensor are the W local CDF sums all-gathered instead.
                   Recovery draws (w_diff > 0): every rank resolves the same draw chain and shard 0 writes
                   the random free-space poses -- no further exchange.

Scoring itself shards with no communication.  The KLD stop rule (an ordered kd-tree replay) runs
redundantly on every rank from the assembled key window, so all ranks agree on the sample count
without another exchange; the resampled set is re-split evenly in index order.

`ShardedFilter` is backend-agnostic: `HipShardBackend` drives the C-ABI stage functions
(bpf_shard_*); the CPU tests plug in a backend of their own over gloo.
"""
import ctypes as C

import numpy as np
import torch


class _DevArray:
    """Zero-copy view of engine-owned device memory for torch (cuda array interface v2)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class HipShardBackend:
    """Stage functions of include/badger_pf.h on one GPU.  Runs on torch's current stream so that
    torch copies / RCCL collectives and engine kernels are ordered without host syncs."""

    def __init__(self, engine, scanner, pf, device):
        self.e, self.sc, self.pf = engine, scanner, pf
        self.device = device
        stream = torch.cuda.current_stream(device)
        engine.set_stream(stream.cuda_stream)
        p = C.c_void_p()
        engine.check(engine.lib.bpf_shard_scalars_dev(engine.h, C.byref(p)))
        self.scalars = torch.as_tensor(_DevArray(p.value, (16,), "<f8"), device=device)
        self._total_view, self._sum_view = self.scalars[0:1], self.scalars[7:8]
        self._engine_device_min = None

    def n_local(self):
        return self.pf.getState().sample_count

    # ---- mailbox exchange (include/badger_pf.h, bpf_shard_mailbox_*): peer stores over xGMI instead of collectives
    def mailbox_create(self, rank, world, max_window):
        """64-byte IPC handle of this engine's mailbox, or None when it cannot be allocated / exported."""
        buf = (C.c_ubyte * 64)()
        rc = self.e.lib.bpf_shard_mailbox_create(self.e.h, rank, world, int(max_window), buf)
        self._mb_world, self._mb_stride, self._mb_views = world, int(max_window), {}
        return bytes(buf) if rc == 0 else None

    def mailbox_connect(self, handles):
        """Maps the peers (handles: world x 64 bytes in rank order) and runs one round with all of them."""
        return self.e.lib.bpf_shard_mailbox_connect(self.e.h, C.c_char_p(handles)) == 0

    def mailbox_update_sensor(self, data, global_n):
        """Scoring, exchange of the totals and normalisation in one call; None when beam skipping needs the
        caller's all-reduce in between (the stage functions finish the update then)."""
        rp, ap = data.pointers()
        rc = self.e.lib.bpf_shard_mailbox_update_sensor_planar(self.e.h, rp, ap, data.range_count_, data.range_max_,
                                                               int(global_n))
        if rc == 100:  # BPF_SHARD_NEED_BEAM_COUNTS
            return None
        self.e.check(rc)
        return True

    def mailbox_update_resample(self, flags, global_n, leaf_count, window_hint):
        """updateResample of the shard in one call: (M, leaf_count, bin_count, windows, window_hint)."""
        if self._engine_device_min != self.kld_device_min:
            self.e.set_option(3, int(self.kld_device_min) if self.kld_device_min < (1 << 31) else 0)  # KLD_DEVICE_MIN
            self._engine_device_min = self.kld_device_min
        g, lf, bn, wn, hint = (C.c_int(int(global_n)), C.c_int(int(leaf_count)), C.c_int(0), C.c_int(0),
                               C.c_int(int(window_hint)))
        self.e.check(self.e.lib.bpf_shard_mailbox_update_resample(self.e.h, C.c_void_p(flags.data_ptr()), C.byref(g),
                                                                  C.byref(lf), C.byref(bn), C.byref(wn),
                                                                  C.byref(hint)))
        return g.value, lf.value, bn.value, wn.value, hint.value

    def mailbox_selftest(self, rounds=4):
        """Full window exchanges with a payload every rank verifies (both parities, twice)."""
        return self.e.lib.bpf_shard_mailbox_selftest(self.e.h, rounds) == 0

    def mailbox_destroy(self):
        self.e.lib.bpf_shard_mailbox_destroy(self.e.h)

    def mailbox_set_timeout_ms(self, ms):
        self.e.check(self.e.lib.bpf_shard_mailbox_set_timeout_ms(self.e.h, int(ms)))

    def mailbox_error_stage(self):
        """(totals wait failed, window wait failed) of the mailbox in use; drains the stream first."""
        a, b = C.c_int(), C.c_int()
        self.e.check(self.e.lib.bpf_shard_mailbox_error_stage(self.e.h, C.byref(a), C.byref(b)))
        return bool(a.value), bool(b.value)

    def _mb_view(self, ptr, shape, typestr):
        v = self._mb_views.get(ptr)
        if v is None:
            v = self._mb_views[ptr] = torch.as_tensor(_DevArray(ptr, shape, typestr), device=self.device)
        return v

    def mailbox_totals(self):
        """The W totals of the scoring stage just issued (complete once normalize has run: it waits in-kernel)."""
        p = C.c_void_p()
        self.e.check(self.e.lib.bpf_shard_mailbox_totals(self.e.h, C.byref(p)))
        return self._mb_view(p.value, (self._mb_world,), "<f8")

    def mailbox_window(self):
        """A fresh [6, max_window] int64 window; valid until the next-but-one call."""
        p, stride = C.c_void_p(), C.c_int()
        self.e.check(self.e.lib.bpf_shard_mailbox_window(self.e.h, C.byref(p), C.byref(stride)))
        return self._mb_view(p.value, (6, stride.value), "<i8")

    def score(self, data):
        lib, e = self.e.lib, self.e
        if hasattr(data, "points_"):  # PointCloudData: the 3-D path
            e.check(lib.bpf_shard_score_cloud(e.h, data.points_.ctypes.data_as(C.POINTER(C.c_float)),
                                              data.points_.shape[0]))
            return
        rp, ap = data.pointers()
        rc = lib.bpf_shard_score_planar(e.h, rp, ap, data.range_count_, data.range_max_)
        if rc != 100:  # BPF_SHARD_NEED_BEAM_COUNTS
            e.check(rc)
            return None
        # beam skipping: the per-beam agreement counts have to be summed over the shards first
        p, n = C.c_void_p(), C.c_int()
        e.check(lib.bpf_shard_beam_counts_dev(e.h, C.byref(p), C.byref(n)))
        return torch.as_tensor(_DevArray(p.value, (n.value,), "<i4"), device=self.device)

    def beam_counts(self):
        p, n = C.c_void_p(), C.c_int()
        self.e.check(self.e.lib.bpf_shard_beam_counts_dev(self.e.h, C.byref(p), C.byref(n)))
        return torch.as_tensor(_DevArray(p.value, (n.value,), "<i4"), device=self.device)

    def score_finish(self, data, global_n):
        e = self.e
        e.check(e.lib.bpf_shard_score_planar_finish(e.h, data.ranges_.ctypes.data_as(C.POINTER(C.c_double)),
                                                    data.angles_.ctypes.data_as(C.POINTER(C.c_double)),
                                                    data.range_count_, data.range_max_, int(global_n)))

    def local_total(self):
        return self._total_view

    def normalize(self, totals, global_n):
        e = self.e
        e.check(e.lib.bpf_shard_normalize_dev(e.h, C.c_void_p(totals.data_ptr()), totals.numel(), int(global_n)))

    def build_cdf(self, flags):
        self.e.check(self.e.lib.bpf_shard_build_cdf(self.e.h, C.c_void_p(flags.data_ptr())))

    def local_sum(self):
        return self._sum_view

    def draw_window(self, rng, m0, m1, sums, sums_are_totals, rank, world, window, flags):
        e = self.e
        e.check(e.lib.bpf_shard_draw_window_dev(e.h, C.c_uint64(rng), m0, m1, C.c_void_p(sums.data_ptr()),
                                                int(sums_are_totals), rank, world, C.c_void_p(window.data_ptr()),
                                                window.shape[1], C.c_void_p(flags.data_ptr())))

    def tail_small(self, x_all, y_all, th_all, m, lo, hi, leaf, bins):
        e = self.e
        e.check(e.lib.bpf_shard_tail_small_dev(e.h, C.c_void_p(x_all.data_ptr()), C.c_void_p(y_all.data_ptr()),
                                               C.c_void_p(th_all.data_ptr()), m, lo, hi, leaf, bins))

    def kld_reset(self):
        self.e.check(self.e.lib.bpf_kld_reset(self.e.h))

    def kld_insert(self, keys_cpu, n):
        k = keys_cpu.numpy()
        assert k.dtype == np.int64 and k.flags.c_contiguous
        self.e.check(self.e.lib.bpf_kld_insert(self.e.h, k.ctypes.data_as(C.c_void_p), 1, k.shape[1], n))

    def kld_feed(self, keys_cpu, n, first):
        stop = C.c_int(-1)
        k = keys_cpu.numpy()
        assert k.dtype == np.int64 and k.flags.c_contiguous
        self.e.check(self.e.lib.bpf_kld_feed(self.e.h, k.ctypes.data_as(C.c_void_p), 1, k.shape[1], n, first,
                                             C.byref(stop)))
        return stop.value

    def kld_feed_window(self, window, n, first):
        """Keys = rows 3..5 of the assembled device window; no torch copy, no stream synchronisation."""
        stop = C.c_int(-1)
        self.e.check(self.e.lib.bpf_kld_feed_dev(self.e.h, C.c_void_p(window.data_ptr()), window.shape[1], n, first,
                                                 C.byref(stop)))
        return stop.value

    def resample_model(self):
        return self.pf.resample_model

    def begin_resample(self, rng, leaf_count):
        """(w_diff, systematic count); resolves the draw chain when w_diff > 0 (multinomial)."""
        w, c = C.c_double(), C.c_int()
        self.e.check(self.e.lib.bpf_shard_begin_resample(self.e.h, C.c_uint64(rng), int(leaf_count), C.byref(w),
                                                         C.byref(c)))
        return w.value, c.value

    def end_resample(self, m):
        """drand48 state after m samples; resets the averages when w_diff > 0."""
        out = C.c_uint64()
        self.e.check(self.e.lib.bpf_shard_end_resample(self.e.h, int(m), C.byref(out)))
        return out.value

    def resample_limit(self, leaf_count):
        out = C.c_int()
        self.e.check(self.e.lib.bpf_pf_resample_limit(self.e.h, int(leaf_count), C.byref(out)))
        return out.value

    def systematic_window(self, rng, count, sums, sums_are_totals, rank, world, window, flags):
        e = self.e
        e.check(e.lib.bpf_shard_systematic_window_dev(e.h, C.c_uint64(rng), count, C.c_void_p(sums.data_ptr()),
                                                      int(sums_are_totals), rank, world,
                                                      C.c_void_p(window.data_ptr()), window.shape[1],
                                                      C.c_void_p(flags.data_ptr())))

    def kld_insert_window(self, window, n):
        self.e.check(self.e.lib.bpf_kld_insert_dev(self.e.h, C.c_void_p(window.data_ptr()), window.shape[1], n))

    def local_pose_keys(self):
        """int64 [3, n_local] histogram keys of the local poses (floor(pose / cell), pf_kdtree.cpp:52-54)."""
        s = self.pf.getCurrentSet().samples
        k = np.stack([np.floor(s[:, 0] / 0.5), np.floor(s[:, 1] / 0.5), np.floor(s[:, 2] / (10 * np.pi / 180))])
        return torch.from_numpy(k.astype(np.int64)).to(self.device)

    kld_device_min = 8192  # draws left after the first window from which the device tree takes the whole stream

    def kld_stop_window(self, window, n):
        """Stop rule for the whole stream on the device: (handled, stop or -1, leaf_count, bin_count)."""
        h, stop, leaf, bins = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.e.check(self.e.lib.bpf_kld_stop_dev(self.e.h, C.c_void_p(window.data_ptr()), window.shape[1], n,
                                                 C.byref(h), C.byref(stop), C.byref(leaf), C.byref(bins)))
        return bool(h.value), stop.value, leaf.value, bins.value

    def kld_counts(self):
        a, b = C.c_int(), C.c_int()
        self.e.check(self.e.lib.bpf_kld_leaf_count(self.e.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def adopt(self, x, y, th, count, global_m, leaf, bins):
        e = self.e
        e.check(e.lib.bpf_shard_adopt_dev(e.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                          C.c_void_p(th.data_ptr()), count, global_m, leaf, bins))

    def converged(self, x_all, y_all, m):
        e = self.e
        e.check(e.lib.bpf_shard_converged_dev(e.h, C.c_void_p(x_all.data_ptr()), C.c_void_p(y_all.data_ptr()), m))

    def skip(self, state, n):
        return int(self.e.lib.bpf_drand48_skip(C.c_uint64(state), C.c_uint64(n)))

    def rng_state(self):
        return self.pf.getRngState()

    def set_rng_state(self, s):
        self.pf.setRngState(s)

    def update_action(self, odom, data, global_first, global_count):
        odom.updateActionShard(data, global_first, global_count)

    def max_samples(self):
        return self.pf.max_samples

    def max_beams(self):
        return getattr(self.sc, "max_beams", 2)

    def state(self):
        return self.pf.getState()


class ShardedState:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class ShardedFilter:
    """ParticleFilter::updateSensor / updateResample over W shards (see module docstring)."""

    def __init__(self, backend, dist, rank=None, world=None, first_window=4096, exchange="auto",
                 mailbox_timeout_ms=None):
        self.b = backend
        self.mailbox_timeout_ms = mailbox_timeout_ms
        self.recoveries = 0  # exchanges that ran out of time and were finished over the collectives
        self._step = 0       # resamples completed (the ranks compare it when they recover)
        self.dist = dist
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.device = backend.device
        self.cpu_collectives = dist.get_backend() == "gloo" and torch.device(self.device).type != "cpu"
        self._nccl = not self.cpu_collectives and dist.get_backend() == "nccl"
        self._gather_out = {}
        self._pose_views = {}
        self.max_global = backend.max_samples()  # engines are created with the GLOBAL min / max sample counts
        n = torch.tensor([backend.n_local()], dtype=torch.int64, device=self.device)
        self.counts = [int(v) for v in self._all_gather(n).cpu().tolist()]
        self.window_hint = first_window
        self.out = torch.zeros((3, self.max_global), dtype=torch.float64, device=self.device)
        self.flags = torch.zeros(4, dtype=torch.int32, device=self.device)
        self._windows = {}
        self.sample_count = sum(self.counts)
        self.leaf_count = self.bin_count = 0
        self.windows_used = 0
        self.totals = None  # per-shard weight totals of the last update_sensor (None: weights changed since)
        self._fused_totals = False  # the last update_sensor went through the engine's one-call mailbox form
        # exchange: "mailbox" = peer stores through IPC-mapped device memory (all ranks on one node), "collective" =
        # torch.distributed all-gather / all-reduce, "auto" = mailbox when every rank could set it up
        self.mailbox = False
        # what the bring-up found: "pass" (every rank created, mapped, completed the connect round and verified the
        # four self-test windows cell by cell), "fail: <stage>" (some rank did not; the collectives carry the
        # exchanges), "not run" (collectives requested)
        self.mailbox_verdict = "not run"
        if exchange not in ("auto", "mailbox", "collective"):
            raise ValueError("exchange: auto, mailbox or collective")
        if exchange != "collective" and hasattr(backend, "mailbox_create") and self.world <= 16:
            self.mailbox = self._setup_mailbox()
        if exchange == "mailbox" and not self.mailbox:
            raise RuntimeError("mailbox exchange requested but not every rank could set it up")
        if backend.resample_model() == 1:
            # the systematic resampler sizes the new set from the leaf count of the CURRENT set's tree, which the
            # reference builds when the set is created (not after motion updates): take it now
            self._global_leaf_count()

    def _all_agree(self, ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        if self.cpu_collectives:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.MIN)
            return int(h.item()) == 1
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item()) == 1

    def _setup_mailbox(self):
        """True when every rank created its mailbox, mapped all the others and completed a round with them."""
        b = self.b
        if self.mailbox_timeout_ms is not None and hasattr(b, "mailbox_set_timeout_ms"):
            b.mailbox_set_timeout_ms(self.mailbox_timeout_ms)
        handle = b.mailbox_create(self.rank, self.world, self.max_global)
        mine = torch.tensor(list(handle if handle is not None else bytes(64)), dtype=torch.uint8, device=self.device)
        handles = self._all_gather(mine).cpu().numpy().tobytes()
        stage = "create / IPC export"
        ok = self._all_agree(handle is not None)
        if ok:
            stage = "IPC map + connect round"
            ok = self._all_agree(b.mailbox_connect(handles))
        if ok:
            # the words arrive; do the window cells?  (a peer's stores must be visible behind this GPU's caches)
            stage = "window self-test"
            ok = self._all_agree(b.mailbox_selftest())
        if not ok:
            b.mailbox_destroy()
        self.mailbox_verdict = "pass" if ok else "fail: " + stage
        return ok

    def use_collectives(self):
        """Drop the mailbox and carry the exchanges over torch.distributed from the next update on (bench.py times
        the same filter once per exchange; every rank calls this between two steps)."""
        if self.mailbox:
            self.b.mailbox_destroy()
        self.mailbox = False
        self._windows.clear()
        self._pose_views.clear()
        self.totals = None
        self._fused_totals = False

    def try_mailbox(self):
        """(Re-)establish the mailbox between two steps; True when every rank could."""
        if not self.mailbox and hasattr(self.b, "mailbox_create") and self.world <= 16:
            self._windows.clear()
            self._pose_views.clear()
            self.totals = None
            self._fused_totals = False
            self.mailbox = self._setup_mailbox()
        return self.mailbox

    # ---- a mailbox wait ran out of time (a rank stalled: page-in, debugger, a long host pause)
    def _is_exchange_error(self, err):
        return getattr(err, "code", None) == 9  # BPF_ERR_EXCHANGE

    def _recover_exchange(self):
        """Every rank gets here after its OWN wait has run out (the rank that stalled finds its peers gone one
        exchange later), so the collectives below are reached by all of them.  The engine's rule (badger_pf.h): after a
        failed wait for the totals the weights are scored but NOT normalised and the local total is in the scalars;
        a failed window wait has changed nothing of the current set.  So: drop the mailbox, all-gather the local
        totals, normalise where that was still due, and go on with the collectives -- the interrupted resample is
        simply run again over them.  The mailbox is set up afresh after the step."""
        b = self.b
        totals_failed, _ = b.mailbox_error_stage()
        # meeting point of the ranks, and a check that they are recovering the same step (a time-out that fell
        # within microseconds of the awaited word can leave them a step apart: that is reported, not papered over)
        steps = self._all_gather(torch.tensor([self._step], dtype=torch.int64, device=self.device)).cpu().tolist()
        if len(set(int(v) for v in steps)) != 1:
            raise RuntimeError("sharded filter: the ranks fell out of step around a mailbox time-out: %r" % (steps,))
        b.mailbox_destroy()
        self.mailbox = False
        self._windows.clear()
        self._pose_views.clear()
        totals = self._all_gather(b.local_total()).clone()
        if totals_failed:
            b.normalize(totals, self.sample_count)
        self.totals = totals
        self._fused_totals = False
        self.recoveries += 1
        self._remake_mailbox = True

    # ---- collectives (device tensors with nccl; staged through the host only for gloo + GPU)
    def _all_gather(self, t):
        if self._nccl:
            # one output buffer per shape, reused: the callers consume the result before the next gather of that shape
            key = (t.numel(), t.dtype)
            out = self._gather_out.get(key)
            if out is None:
                out = self._gather_out[key] = torch.empty(self.world * t.numel(), dtype=t.dtype, device=t.device)
            self.dist.all_gather_into_tensor(out, t if t.is_contiguous() else t.contiguous())
            return out
        src = t.cpu() if self.cpu_collectives else t
        outs = [torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(outs, src.contiguous())
        res = torch.cat(outs)
        return res.to(self.device) if self.cpu_collectives else res

    def _all_reduce_sum(self, t):
        if self.cpu_collectives:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def _window(self, key, count):
        """The [6, >= count] int64 draw window of the next exchange."""
        if self.mailbox:
            return self.b.mailbox_window()
        window = self._windows.get(key)
        if window is None:
            window = self._windows[key] = torch.zeros((6, count), dtype=torch.int64, device=self.device)
        return window

    def _assemble(self, window):
        """Every shard's columns into every shard's window: nothing to do with a mailbox (the draw kernel stored
        them into all peers, the window's first consumer waits for them), one integer all-reduce otherwise."""
        if not self.mailbox:
            self._all_reduce_sum(window)

    def _pose_rows(self, window):
        """(x, y, theta) float64 row views of a window buffer; the buffers live in self._windows, so are the views."""
        v = self._pose_views.get(id(window))
        if v is None:
            p = window[0:3].view(torch.float64)
            v = self._pose_views[id(window)] = (p[0], p[1], p[2], window)  # keeps the buffer alive with its id
        return v

    # ---- motion update (Odom::updateAction): no exchange
    def update_action(self, odom, data):
        first = sum(self.counts[:self.rank])
        self.b.update_action(odom, data, first, self.sample_count)

    # ---- Seam A
    def update_sensor(self, data):
        if self.mailbox and not hasattr(data, "points_") and hasattr(self.b, "mailbox_update_sensor"):
            # mailbox: the exchange is inside the kernels, so the whole update is one call into the engine
            if self.b.mailbox_update_sensor(data, self.sample_count):
                if self.b.max_beams() < 2:
                    # PlanarScanner::updateSensor is a no-op then (planar_scanner.cpp:128-129): nothing was posted
                    self.totals = None
                    self._fused_totals = False
                    return
                self.totals = self.b.mailbox_totals()
                self._fused_totals = True
                return
            # beam skipping: the counting pass has run; sum its counts over the shards and finish stage by stage
            counts = self.b.beam_counts()
            self._all_reduce_sum(counts)
            self.b.score_finish(data, self.sample_count)
            self.totals = self.b.mailbox_totals()
            self.b.normalize(self.totals, self.sample_count)
            self._fused_totals = False
            return
        self._fused_totals = False
        counts = self.b.score(data)
        if counts is not None:
            # prob model with beam skipping: one extra all-reduce of max_beams int32 between its two passes
            self._all_reduce_sum(counts)
            self.b.score_finish(data, self.sample_count)
        # mailbox: the scoring stage has already stored this rank's total into every peer, normalize waits in-kernel
        self.totals = self.b.mailbox_totals() if self.mailbox else self._all_gather(self.b.local_total())
        self.b.normalize(self.totals, self.sample_count)

    def _global_leaf_count(self):
        """Leaf count of the kd-tree of the whole current set (what set_a->kdtree->getLeafCount() is for the
        reference's systematic resampler): known after a resample, otherwise built once from all shards' keys."""
        if self.leaf_count > 0:
            return self.leaf_count
        b = self.b
        pad = max(self.counts)
        mine = torch.zeros((3, pad), dtype=torch.int64, device=self.device)
        k = b.local_pose_keys()
        mine[:, :k.shape[1]] = k
        allk = self._all_gather(mine.reshape(-1)).reshape(self.world, 3, pad).cpu()
        b.kld_reset()
        for r in range(self.world):
            if self.counts[r]:
                b.kld_insert(allk[r, :, :self.counts[r]].contiguous(), self.counts[r])
        self.leaf_count, self.bin_count = b.kld_counts()
        return self.leaf_count

    # ---- Seam B, systematic (particle_filter.cpp:269-354, w_diff == 0)
    def _update_resample_systematic(self):
        b, W = self.b, self.world
        rng = b.rng_state()
        w_diff, count = b.begin_resample(rng, self._global_leaf_count())
        b.build_cdf(self.flags)
        if self.totals is not None:
            sums, sums_are_totals = self.totals, True
        else:
            sums, sums_are_totals = self._all_gather(b.local_sum()), False
        window = self._window((count, "sys"), count)
        b.systematic_window(rng, count, sums, sums_are_totals, self.rank, W, window, self.flags)
        self._assemble(window)
        b.kld_reset()
        b.kld_insert_window(window, count)  # the tree of the new set: every sample, no stop rule
        leaf, bins = b.kld_counts()
        M = count
        lo, hi = (M * self.rank) // W, (M * (self.rank + 1)) // W
        pose = self._pose_rows(window)
        if M <= 8192:
            b.tail_small(pose[0], pose[1], pose[2], M, lo, hi, leaf, bins)
        else:
            b.adopt(pose[0][lo:hi], pose[1][lo:hi], pose[2][lo:hi], hi - lo, M, leaf, bins)
            b.converged(pose[0][:M], pose[1][:M], M)
        b.set_rng_state(b.end_resample(M))
        self.counts = [(M * (r + 1)) // W - (M * r) // W for r in range(W)]
        self.sample_count = M
        self.leaf_count, self.bin_count = leaf, bins
        self.windows_used = 1
        self.totals = None

    # ---- Seam B (multinomial, w_diff == 0)
    def update_resample(self):
        self._update_resample()
        self._step += 1

    def _update_resample(self):
        b, W = self.b, self.world
        if self.mailbox and getattr(self, "_fused_totals", False) and hasattr(b, "mailbox_update_resample"):
            # one call: CDF, windows, stop rule, adoption of this rank's share (bpf_shard_mailbox_update_resample)
            leaf_in = self._global_leaf_count() if b.resample_model() == 1 else self.leaf_count
            try:
                M, leaf, bins, wins, hint = b.mailbox_update_resample(self.flags, self.sample_count, leaf_in,
                                                                      self.window_hint)
            except Exception as err:  # noqa: BLE001 -- only the exchange time-out is handled, the rest goes up
                if not self._is_exchange_error(err):
                    raise
                self._recover_exchange()
                self._update_resample_stages()
                self._finish_recovery()
                return
            self.counts = [(M * (r + 1)) // W - (M * r) // W for r in range(W)]
            self.sample_count, self.leaf_count, self.bin_count = M, leaf, bins
            self.windows_used, self.window_hint = wins, hint
            self.totals = None
            self._fused_totals = False
            return
        self._update_resample_stages()

    def _finish_recovery(self):
        if getattr(self, "_remake_mailbox", False):
            self._remake_mailbox = False
            self.mailbox = self._setup_mailbox()

    def _update_resample_stages(self):
        """updateResample stage by stage (collectives, or a mailbox with the host between the stages)."""
        b, W = self.b, self.world
        if b.resample_model() == 1:  # PF_RESAMPLE_SYSTEMATIC
            return self._update_resample_systematic()
        b.build_cdf(self.flags)
        if self.totals is not None:
            sums, sums_are_totals = self.totals, True   # slices from the sensor update's totals
        else:
            sums, sums_are_totals = self._all_gather(b.local_sum()), False
        rng = b.rng_state()
        b.begin_resample(rng, self.leaf_count)  # w_diff; with w_diff > 0 the draws follow the resolved chain
        b.kld_reset()
        m0, stop = 0, -1
        win = max(1024, min(self.window_hint, self.max_global))
        self.windows_used = 0
        windows = []
        device_counts = None
        device_min = b.kld_device_min
        if win > 4096 and self.max_global - 4096 >= device_min:
            win = 4096  # keep the host's first window short when the device tree can take over after it
        need = 0  # after a window without a stop: draws still missing to the bound for the leaves seen so far
        while m0 < self.max_global and stop < 0:
            if m0 > 0 and self.max_global - m0 >= device_min and need >= device_min:
                # no stop in the first window and a long stream ahead (a spread cloud): one window with every
                # remaining candidate, and the ordered kd-tree replay runs on the device (every rank, redundantly)
                whole = self._window("whole", self.max_global)
                b.draw_window(rng, 0, self.max_global, sums, sums_are_totals, self.rank, W, whole, self.flags)
                self._assemble(whole)
                handled, dstop, dleaf, dbins = b.kld_stop_window(whole, self.max_global)
                self.windows_used += 1
                if handled:
                    stop = dstop
                    windows = [(0, self.max_global, whole)]
                    device_counts = (dleaf, dbins)
                    break
                device_min = 1 << 62  # this stream is outside what the device tree takes: host replay
            m1 = min(self.max_global, m0 + win)
            cnt = m1 - m0
            window = self._window((cnt, len(windows)), cnt)
            b.draw_window(rng, m0, m1, sums, sums_are_totals, self.rank, W, window, self.flags)
            self._assemble(window)
            stop = b.kld_feed_window(window, cnt, m0)  # the one host wait of the window
            if self.mailbox and stop < 0:
                # a mailbox window is overwritten two exchanges later: keep its poses now, the stream goes on
                self.out[:, m0:m0 + cnt] = window[0:3, :cnt].view(torch.float64)
                window = None
            windows.append((m0, cnt, window))
            self.windows_used += 1
            m0 = m1
            if stop < 0:
                # next window: up to a quarter past the bound for the leaves seen so far (a lower estimate of the stop)
                need = b.resample_limit(b.kld_counts()[0]) - m0
                win = max(1024, (need + need // 4 + 1023) // 1024 * 1024)
        M = stop if stop > 0 else self.max_global
        leaf, bins = device_counts if device_counts is not None else b.kld_counts()
        lo, hi = (M * self.rank) // W, (M * (self.rank + 1)) // W
        if len(windows) == 1 and M <= 8192 and windows[0][2] is not None:
            # the common case: one window, small set -> adopt + weights + updateConverged in one launch
            pose = self._pose_rows(windows[0][2])
            b.tail_small(pose[0], pose[1], pose[2], M, lo, hi, leaf, bins)
        else:
            for (w0, cnt, window) in windows:
                if window is not None:
                    self.out[:, w0:w0 + cnt] = window[0:3, :cnt].view(torch.float64)
            b.adopt(self.out[0, lo:hi], self.out[1, lo:hi], self.out[2, lo:hi], hi - lo, M, leaf, bins)
            b.converged(self.out[0, :M], self.out[1, :M], M)
        b.set_rng_state(b.end_resample(M))
        self.counts = [(M * (r + 1)) // W - (M * r) // W for r in range(W)]
        self.sample_count = M
        self.leaf_count, self.bin_count = leaf, bins
        self.window_hint = max(1024, ((M + M // 4) + 1023) // 1024 * 1024)
        self.totals = None  # the weights are 1/M now; the old totals no longer describe them

    def restore(self, counts, leaf_count=0):
        """Bench helper: the shards were put back by pf.restore(); reset the bookkeeping."""
        self.counts = list(counts)
        self.sample_count = sum(counts)
        self.totals = None
        self._fused_totals = False
        self.leaf_count = leaf_count

    def state(self):
        st = self.b.state()
        miss = self._all_reduce_sum(self.flags.clone())
        return ShardedState(sample_count=self.sample_count, local_count=st.sample_count, leaf_count=self.leaf_count,
                            bin_count=self.bin_count, converged=st.converged,
                            percent_converged=st.percent_converged, w_slow=st.w_slow, w_fast=st.w_fast,
                            total=st.total, cdf_miss=int(miss[0].item()) != 0, windows=self.windows_used)
