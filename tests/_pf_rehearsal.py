"""TEST INFRASTRUCTURE — a rehearsal harness, not the product path (the product's frame loop is the C session
slam_pf_* in csrc/pf_session.hip, which issues its own exchange steps).  This module drives the STAGE entry points of
the C ABI from Python with torch.distributed between them, so that the sharding / exchange index algebra can run on
CPUs under gloo with the oracle's stages substituted (tests/test_pf_sharding_gloo.py) and against the C session on the GPU
(tests/test_gpu_pf.py).

Particle-filter frame loop on the engine: host-side plumbing only.

One process per GPU.  This module owns the device buffers (torch tensors are used purely as HBM
allocations on the current stream) and the exchange steps between GPUs (``torch.distributed``, i.e.
RCCL over xGMI with the ``nccl`` backend); every arithmetic stage is a call into the C ABI
(``include/slam_hip.h``) through ``HipOps``.  There is no CPU compute path here: ``HipOps`` needs a
gfx950 engine.  (The CPU tests of the sharding logic substitute their own ``ops`` object built on
the checker in ``oracle/`` — test infrastructure, not shipped.)

Frame t (SURVEY.md §8a rows A9-A12 around the reference's scan-match score, row A7):
  1. motion sample     pose'[i] = pose[src(i)] + dp + eps_i           (resample gather fused in)
  2. scan-match score  score[i] = sum_b EDT[cell(pose'[i] (+) beam_b)]  == FastMatch's inner loop,
                                                                       Subsystem_1/main.c:459-518
  3. per-landmark EKF  map'[l][i] = update(map[l][src(i)], z_l)       (gather fused in), loglik[i]
  4. weights           logw = loglik - gain*score ; m = max (all-reduce MAX over GPUs)
                       wq = fixed-point exp(logw - m) ; shard totals all-gathered (8 B per GPU)
  5. resample          integer CDF (wavefront prefix sum) -> first slot of every particle
                       -> all-gather of those indices -> ancestor of every local slot
  6. migration         particles whose ancestor lives on another GPU are fetched with one
                       all-to-all of poses (12 B each) and, when maps exist, of map rows (20 B x L)
Results are bit-identical for any number of GPUs: noise is keyed by global particle id, the CDF is
exact integer arithmetic, and the comb offset is a pure function of (seed, frame, total).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class HipOps:
    """Stage calls -> C ABI.  Tensors must live on the engine's GPU."""

    def __init__(self, engine):
        self.e = engine

    def bind_stream(self):
        self.e.set_stream(torch.cuda.current_stream().cuda_stream)

    def motion_sample(self, src, anc, dst, n, first_id, dp, sigma, seed, frame):
        self.e.motion_sample_dev(src, anc, dst, n, first_id, dp, sigma, seed, frame)

    def score(self, slot, x, y, th, n, score, count):
        self.e.score_poses_dev(slot, x, y, th, n, score, count)

    def motion_score(self, slot, src, anc, dst, n, first_id, dp, sigma, seed, frame, score, count):
        self.e.motion_score_dev(slot, src, anc, dst, n, first_id, dp, sigma, seed, frame, score, count)

    def obs_set_dev(self, d_zx_by_landmark, d_zy_by_landmark, nlandmarks):
        self.e.obs_set_dev(d_zx_by_landmark, d_zy_by_landmark, nlandmarks)

    def logweight_ekf(self, score, gain, n, logw, d_max):
        self.e.logweight_ekf_dev(score, gain, n, logw, d_max)

    def obs_upload(self, ids, zx, zy, nlandmarks):
        self.e.obs_upload(ids, zx, zy, nlandmarks)

    def ekf(self, map_in, map_out, row_stride, plane_stride, nlandmarks, x, y, th, anc, n, meas_var, loglik):
        self.e.ekf_update_dev(map_in, map_out, row_stride, plane_stride, nlandmarks, x, y, th, anc, n, meas_var, loglik)

    def logweight(self, score, loglik, gain, n, logw, d_max):
        self.e.logweight_dev(score, loglik, gain, n, logw, d_max)

    def quantise(self, logw, d_max, n, wq, d_sum):
        self.e.quantise_weights_dev(logw, d_max, n, wq, d_sum)

    def prefix_sum(self, wq, n, cdf):
        self.e.prefix_sum_dev(wq, n, cdf)

    def quantise_scan(self, logw, d_max, n, d_sum):
        self.e.quantise_scan_dev(logw, d_max, n, d_sum)

    def offspring_from_scan(self, n, d_base, d_total, seed, frame, n_total, first):
        self.e.offspring_from_scan_dev(n, d_base, d_total, seed, frame, n_total, first)

    def offspring_from_scan_sharded(self, n, totals, rank, world, seed, frame, n_total, first):
        self.e.offspring_from_scan_sharded_dev(n, totals, rank, world, seed, frame, n_total, first)

    def offspring_offsets(self, cdf, n, d_base, d_total, seed, frame, n_total, first):
        self.e.offspring_offsets_dev(cdf, n, d_base, d_total, seed, frame, n_total, first)

    def ancestors(self, first_all, n_total, slot0, nslots, anc):
        self.e.ancestors_dev(first_all, n_total, slot0, nslots, anc)

    def ancestors_from_scan(self, n, seed, frame, anc):
        self.e.ancestors_from_scan_dev(n, seed, frame, anc)

    def ancestors_sharded(self, first_all, n_total, n_local, rank, world, src, plan, pose_idx=None):
        self.e.ancestors_sharded_dev(first_all, n_total, n_local, rank, world, src, plan, pose_idx)

    def set_exchange_capacity(self, rows):
        self.e.exchange_set_capacity(rows)

    def read_plan(self, d_plan, world):
        return self.e.exchange_plan_host(world)   # zero-copy: the plan kernel wrote it to mapped host memory

    def migrate_pack(self, n_local, rank, world, plan, pose, pose_ld, mp, row_stride, plane_stride, nlandmarks, out):
        self.e.migrate_pack_dev(n_local, rank, world, plan, pose, pose_ld, mp, row_stride, plane_stride, nlandmarks, out)

    def migrate_unpack(self, inp, world, recv_cnt, n_local, pose, pose_ld, mp, row_stride, plane_stride, nlandmarks):
        self.e.migrate_unpack_dev(inp, world, recv_cnt, n_local, pose, pose_ld, mp, row_stride, plane_stride, nlandmarks)

    def gather_f32(self, src, idx, n, dst):
        self.e.gather_f32_dev(src, idx, n, dst)

    def gather_map(self, m_in, m_out, in_row_stride, out_row_stride, in_plane_stride, out_plane_stride, nlandmarks, idx, n):
        self.e.gather_map_dev(m_in, m_out, in_row_stride, out_row_stride, in_plane_stride, out_plane_stride, nlandmarks,
                              idx, n)


class ParticleFilter:
    """Sharded FastSLAM-style filter: ``n_local`` particles on this rank, ``world * n_local`` in total."""

    def __init__(self, ops, n_local: int, nlandmarks: int = 0, *, device, rank: int = 0, world: int = 1, group=None,
                 seed: int = 1, sigma=(0.01, 0.01, 0.002), meas_var: float = 0.01, score_gain: float = 1.0,
                 grid_slot: int = 0, recv_capacity: int | None = None, force_collectives: bool = False):
        self.ops, self.n, self.L = ops, int(n_local), int(nlandmarks)
        self.rank, self.world, self.group = rank, world, group
        # force_collectives: take the multi-GPU code path (every collective, the sharded kernels) even with one
        # rank — lets a single GPU exercise the RCCL calls themselves
        self.multi = world > 1 or force_collectives
        self.n_total = self.n * world
        if self.n_total >= 2**31:
            raise ValueError("total particle count must stay below 2^31 (int32 ancestor indices)")
        self.device = torch.device(device)
        self.seed, self.sigma, self.meas_var, self.score_gain = seed, tuple(sigma), meas_var, score_gain
        self.grid_slot = grid_slot
        self.recv_cap = 0 if not self.multi else (self.n if recv_capacity is None else int(recv_capacity))
        self.cap = self.n + self.recv_cap
        dv = self.device
        f32, i32, i64 = torch.float32, torch.int32, torch.int64
        self.pose = torch.zeros((2, 3, self.n), dtype=f32, device=dv)           # [buffer][x,y,theta][particle]
        # landmark maps: one row per particle, [buffer][particle][plane][Lp]; planes padded to 32 floats so that
        # every row starts on a 128-byte boundary (include/slam_hip.h, slam_ekf_update_dev)
        self.Lp = (self.L + 31) // 32 * 32
        self.map = torch.zeros((2, self.cap, 5, self.Lp), dtype=f32, device=dv) if self.L else None
        self.cur = 0
        self.src_idx = None            # local gather indices left by the previous resample (None = identity)
        self.score = torch.zeros(self.n, dtype=f32, device=dv)
        self.count = torch.zeros(self.n, dtype=i32, device=dv)
        self.loglik = torch.zeros(self.n, dtype=f32, device=dv)
        self.logw = torch.zeros(self.n, dtype=f32, device=dv)
        self.first = torch.zeros(self.n, dtype=i32, device=dv)
        self.first_all = self.first if not self.multi else torch.zeros(self.n_total, dtype=i32, device=dv)
        self.anc = torch.zeros((2, self.n), dtype=i32, device=dv)                # double-buffered: the fused gathers
        self.d_max = torch.zeros(1, dtype=f32, device=dv)                         # of frame t+1 read frame t's indices
        self.d_sum = torch.zeros(1, dtype=i64, device=dv)
        self.totals = torch.zeros(world, dtype=i64, device=dv)
        self.plan = torch.zeros(1 + 3 * world, dtype=i32, device=dv)         # exchange plan of the frame (slam_hip.h)
        # several GPUs: the poses of every rank are all-gathered each frame (12 B per particle, overlapped with the
        # EKF), so that the next frame's motion + score launch does not have to wait for the exchange of map rows
        self.pose_all = torch.zeros(3 * self.n_total, dtype=f32, device=dv) if self.multi else None   # [rank][x|y|th][n]
        self.pose_stage = torch.zeros((3, self.cap), dtype=f32, device=dv) if self.multi else None    # unpack target
        self.pose_idx = torch.zeros((2, self.n), dtype=i32, device=dv) if self.multi else None
        self._pose_work = None           # async all-gather of the poses in flight
        self._exchange_pending = False   # resample done, map rows not exchanged yet
        # views used every frame, made once (slicing a tensor costs a few microseconds of host time each)
        self._rows = [tuple(self.pose[b][k] for k in range(3)) for b in range(2)]
        self._flat = [self.pose[b].reshape(-1) for b in range(2)]
        if self.multi:
            pa, n_ = self.pose_all, self.n
            self._all_rows = (pa, pa[n_:], pa[2 * n_:])
            self._pidx = [self.pose_idx[b] for b in range(2)]
        self._anc = [self.anc[b] for b in range(2)]
        self.frame = 0
        self.migrated_last = 0
        # gloo cannot move GPU tensors for every collective used here: stage them through the host then
        # (functional rehearsal of the multi-rank path on one card; the production backend is nccl = RCCL)
        self._host_staged = self.multi and self.device.type == "cuda" and dist.get_backend(group) == "gloo"
        if hasattr(ops, "bind_stream"):
            ops.bind_stream()
        if self.multi:
            ops.set_exchange_capacity(self.recv_cap)   # every plan then says whether ANY rank's staging area could overflow

    # ------------------------------------------------------------------ collectives (plumbing only)
    def _all_reduce_max(self, t):
        if self._host_staged:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)

    def _all_gather(self, out, t, async_op=False):
        if self._host_staged:
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h, t.cpu(), group=self.group)
            out.copy_(h)
            return None
        return dist.all_gather_into_tensor(out, t, group=self.group, async_op=async_op)

    def _all_to_all(self, out, inp, out_splits, in_splits):
        if self._host_staged:
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(h, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(h)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    # ------------------------------------------------------------------ state access
    def _drop_resample(self):
        """set_poses / set_map discard the pending resample gather: nothing of it may run later."""
        self.src_idx = None
        self._exchange_pending = False
        if self._pose_work is not None:
            self._pose_work.wait()
            self._pose_work = None

    def set_poses(self, x, y, th):
        for k, a in enumerate((x, y, th)):
            self.pose[self.cur, k, : self.n] = torch.as_tensor(a, dtype=torch.float32).to(self.device)
        self._drop_resample()

    def set_map(self, rows):
        """rows: float32 [n_local][5][L] (mu_x, mu_y, P_xx, P_xy, P_yy per landmark)"""
        self.map[self.cur, : self.n, :, : self.L] = torch.as_tensor(rows, dtype=torch.float32).to(self.device)
        self._drop_resample()

    def poses(self):
        """Current particle poses [3][n_local] with any pending resample gather applied.  (Several GPUs: this and
        maps() may finish a pending exchange, a collective — call them on every rank or on none.)"""
        p = self.pose[self.cur]
        if self.src_idx is None:
            return p
        if not self.multi:
            return p[:, self.src_idx.long()]
        if self._pose_work is not None:   # the ancestors' poses are in the all-gathered array
            self._pose_work.wait()
            self._pose_work = None
        idx, n = self.pose_idx[self.cur].long(), self.n
        return torch.stack([self.pose_all[idx], self.pose_all[n + idx], self.pose_all[2 * n + idx]])

    def maps(self):
        """Current maps [n_local][5][L] with any pending resample gather applied."""
        self._finish_exchange()
        m = self.map[self.cur][:, :, : self.L]
        return m[: self.n] if self.src_idx is None else m[self.src_idx.long()]

    # ------------------------------------------------------------------ one frame
    def step(self, dp, obs=None, obs_dev=None):
        """dp: odometry increment (3 floats).  Observations of this frame, either
        obs = (landmark ids, zx, zy) host arrays (uploaded here), or
        obs_dev = (d_zx_by_landmark, d_zy_by_landmark): the table form already resident on the device — entry l is
        the observation of landmark l, NaN in zx = not observed this frame."""
        o, n, cur, nxt = self.ops, self.n, self.cur, 1 - self.cur
        src, dst = self._rows[cur], self._rows[nxt]
        multi = self.multi
        # 1+2. motion (+ fused gather of the previous resample) and scan-match score, one launch.  Several GPUs: the
        # ancestors' poses come out of the all-gathered pose array, so this launch needs nothing from the exchange
        # below and keeps the GPU busy while the host picks up the exchange plan.
        if multi and self.src_idx is not None:
            if self._pose_work is not None:
                self._pose_work.wait()
                self._pose_work = None
            o.motion_score(self.grid_slot, self._all_rows, self._pidx[cur], dst, n,
                           self.rank * n, dp, self.sigma, self.seed, self.frame, self.score, self.count)
        else:
            o.motion_score(self.grid_slot, src, self.src_idx, dst, n,
                           self.rank * n, dp, self.sigma, self.seed, self.frame, self.score, self.count)
        if multi:
            # map rows of remote ancestors -> staging tail; issued behind the launch above, which does not need them
            self._finish_exchange()
            # this frame's poses to every rank for the next frame's motion + score.  Collectives of one communicator run
            # in issue order: after the exchange (the EKF waits for that one), before this frame's all-reduce (so that it
            # runs beside the EKF, not in front of the weight normaliser).
            self._pose_work = self._all_gather(self.pose_all, self._flat[nxt], async_op=True)
        # 3. per-landmark EKF (+ fused gather); the log-likelihood stays inside the engine for step 4
        use_ll = self.L > 0 and (obs is not None or obs_dev is not None)
        if use_ll:
            if obs_dev is not None:
                o.obs_set_dev(*obs_dev, self.L)
            else:
                o.obs_upload(obs[0], obs[1], obs[2], self.L)
            o.ekf(self.map[cur], self.map[nxt], 5 * self.Lp, self.Lp, self.L, dst[0], dst[1], dst[2],
                  self.src_idx, n, self.meas_var, None)
        elif self.L > 0:   # no observation this frame: the maps still have to follow their particles
            idx = self.src_idx if self.src_idx is not None else torch.arange(n, dtype=torch.int32, device=self.device)
            o.gather_map(self.map[cur], self.map[nxt], 5 * self.Lp, 5 * self.Lp, self.Lp, self.Lp, self.L, idx, n)
        # 4. weights (fused form: the fixed-point weights are scanned as they are produced, never stored)
        if use_ll:
            o.logweight_ekf(self.score, self.score_gain, n, self.logw, self.d_max if multi else None)
        else:
            o.logweight(self.score, None, self.score_gain, n, self.logw, self.d_max if multi else None)
        if multi:
            self._all_reduce_max(self.d_max)
        o.quantise_scan(self.logw, self.d_max if multi else None, n, self.d_sum if multi else None)
        # 5. resample on the integer CDF
        if multi:
            self._all_gather(self.totals, self.d_sum)
            o.offspring_from_scan_sharded(n, self.totals, self.rank, self.world, self.seed, self.frame, self.n_total,
                                          self.first)
        anc = self._anc[nxt]
        self.cur = nxt
        if not multi:
            o.ancestors_from_scan(n, self.seed, self.frame, anc)   # offspring offsets + ancestors in one launch
            self.migrated_last = 0
        else:
            # 6. particles whose ancestor lives on another GPU.  The gather index and the exchange plan are made on
            # the device; the host reads the plan once (3 * world + 1 words) for the all-to-all's split sizes.
            self._all_gather(self.first_all, self.first)
            o.ancestors_sharded(self.first_all, self.n_total, n, self.rank, self.world, anc, self.plan, self._pidx[nxt])
            self._exchange_pending = True   # done at the start of the next frame, behind its motion + score launch
        self.src_idx = anc
        self.frame += 1

    # ------------------------------------------------------------------ multi-GPU exchange
    def _finish_exchange(self):
        if self._exchange_pending:
            self._exchange_pending = False
            self._migrate()

    def _migrate(self):
        """pack (one launch) -> one all-to-all carrying poses and map rows -> unpack (one launch) into the
        staging tail of the current buffers, where the next frame's fused gathers pick them up.  A remote ancestor
        travels once per destination rank, however many slots there descend from it."""
        o, n, r, G, dv, L = self.ops, self.n, self.rank, self.world, self.device, self.L
        plan = o.read_plan(self.plan, G)   # the one point of a frame where the host waits for the device
        anything, scnt, rcnt = plan[0] & 1, plan[1:1 + G], plan[1 + G:1 + 2 * G]
        if plan[0] & 2:   # the same verdict on every rank (derived from the all-gathered offsets): all raise, nobody hangs
            raise RuntimeError(f"rank {r}: the exchange of this frame might exceed recv_capacity {self.recv_cap} on some rank")
        if not anything and self.world > 1:   # every run boundary coincides with a rank boundary: all ranks skip
            self.migrated_last = 0
            return
        stot, rtot = sum(scnt), sum(rcnt)
        self.migrated_last = rtot
        rows = 3 + 5 * L
        sbuf = self._exchange_buffer("_sbuf", rows * stot)
        rbuf = self._exchange_buffer("_rbuf", rows * rtot)
        pose = self.pose[self.cur]
        mp = self.map[self.cur] if L else None
        if stot:
            o.migrate_pack(n, r, G, plan, pose, n, mp, 5 * self.Lp, self.Lp, L, sbuf)
        self._all_to_all(rbuf, sbuf, [rows * c for c in rcnt], [rows * c for c in scnt])
        # the records carry the poses too; nothing reads them from here (the next frame takes poses from the
        # all-gathered array), they land in a scratch staging area
        if rtot:
            o.migrate_unpack(rbuf, G, rcnt, n, self.pose_stage, self.cap, mp, 5 * self.Lp, self.Lp, L)

    def _exchange_buffer(self, name, nfloats):
        """Grow-only device buffer for the all-to-all (no allocator traffic in the frame loop)."""
        buf = getattr(self, name, None)
        if buf is None or buf.numel() < nfloats:
            buf = torch.empty(max(nfloats, 2 * (buf.numel() if buf is not None else 0), 1), dtype=torch.float32,
                              device=self.device)
            setattr(self, name, buf)
        return buf[:nfloats]

    # ------------------------------------------------------------------ estimate
    def best_particle(self):
        """(log-weight, global id) of the heaviest particle of the last frame (lowest id on ties)."""
        lw, i = torch.max(self.logw, dim=0)
        cand = torch.stack([lw.double(), (self.rank * self.n + i).double()])
        if self.world > 1:
            allc = torch.empty((self.world, 2), dtype=torch.float64, device=self.device)
            self._all_gather(allc, cand.reshape(1, 2))
            k = int(torch.argmax(allc[:, 0]))   # first max = lowest rank = lowest id
            cand = allc[k]
        return float(cand[0]), int(cand[1])
