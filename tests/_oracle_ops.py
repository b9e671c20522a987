"""`ops` object for pf.ParticleFilter built on the CPU checker (oracle/): TEST INFRASTRUCTURE.

Lets the sharding / exchange logic of the frame loop run on CPU tensors with the gloo backend
(world_size 2) so that it can be compared with an unsharded run — without a GPU.  Never shipped,
never used by the product path."""
import ctypes as C

import numpy as np
import torch

import oracle


def _np(t):
    return None if t is None else t.numpy()


class OracleOps:
    def __init__(self, grid_meta, edt, bx, by):
        self.meta, self.edt, self.bx, self.by = grid_meta, np.ascontiguousarray(edt, np.float32), bx, by
        self.obs = None
        self.exch_cap = None

    def set_exchange_capacity(self, rows):
        self.exch_cap = rows if rows and rows > 0 else None

    def motion_sample(self, src, anc, dst, n, first_id, dp, sigma, seed, frame):
        x, y, th = oracle.motion_sample(_np(src[0]), _np(src[1]), _np(src[2]), _np(anc), n, first_id, dp, sigma, seed, frame)
        dst[0][:n] = torch.from_numpy(x); dst[1][:n] = torch.from_numpy(y); dst[2][:n] = torch.from_numpy(th)

    def score(self, slot, x, y, th, n, score, count):
        s, c = oracle.score_poses_det(self.meta, self.edt, self.bx, self.by, _np(x)[:n], _np(y)[:n], _np(th)[:n])
        score[:n] = torch.from_numpy(s); count[:n] = torch.from_numpy(c)

    def motion_score(self, slot, src, anc, dst, n, first_id, dp, sigma, seed, frame, score, count):
        self.motion_sample(src, anc, dst, n, first_id, dp, sigma, seed, frame)
        self.score(slot, dst[0], dst[1], dst[2], n, score, count)

    def obs_set_dev(self, d_zx_by_landmark, d_zy_by_landmark, nlandmarks):
        zx, zy = _np(d_zx_by_landmark)[:nlandmarks], _np(d_zy_by_landmark)[:nlandmarks]
        ids = np.flatnonzero(~np.isnan(zx)).astype(np.int32)
        self.obs = (ids, zx[ids].copy(), zy[ids].copy())

    def logweight_ekf(self, score, gain, n, logw, d_max):
        self.logweight(score, torch.from_numpy(self._ll), gain, n, logw, d_max)

    def obs_upload(self, ids, zx, zy, nlandmarks):
        self.obs = (np.asarray(ids, np.int32), np.asarray(zx, np.float32), np.asarray(zy, np.float32))

    def ekf(self, map_in, map_out, row_stride, plane_stride, nlandmarks, x, y, th, anc, n, meas_var, loglik):
        L = oracle.lib()
        ll = np.empty(n, np.float32)
        a = _np(anc)
        L.orc_ekf_update(_np(map_in), _np(map_out), row_stride, plane_stride,
                         nlandmarks, _np(x), _np(y), _np(th), a.ctypes.data_as(C.c_void_p) if a is not None else None, n,
                         self.obs[0], self.obs[1], self.obs[2], len(self.obs[0]), meas_var, ll)
        self._ll = ll
        if loglik is not None:
            loglik[:n] = torch.from_numpy(ll)

    def logweight(self, score, loglik, gain, n, logw, d_max):
        lw, m = oracle.logweight(_np(score), _np(loglik), gain)
        logw[:n] = torch.from_numpy(lw)
        if d_max is not None:
            d_max[0] = float(m)

    def quantise(self, logw, d_max, n, wq, d_sum):
        q, s = oracle.quantise_weights(_np(logw), float(d_max[0]))
        wq[:n] = torch.from_numpy(q.view(np.int64)); d_sum[0] = s

    def prefix_sum(self, wq, n, cdf):
        cdf[:n] = torch.from_numpy(oracle.prefix_sum(_np(wq).view(np.uint64)).view(np.int64))

    def quantise_scan(self, logw, d_max, n, d_sum):
        lw = _np(logw)[:n]
        m = float(d_max[0]) if d_max is not None else float(lw.max())
        q, s = oracle.quantise_weights(lw, m)
        self._cdf = oracle.prefix_sum(q)
        if d_sum is not None:
            d_sum[0] = s

    def offspring_from_scan(self, n, d_base, d_total, seed, frame, n_total, first):
        base = int(d_base[0]) if d_base is not None else 0
        total = int(d_total[0]) if d_total is not None else int(self._cdf[-1])
        u = oracle.comb_offset(seed, frame, total)
        first[:n] = torch.from_numpy(oracle.offspring_offsets(self._cdf, base, total, u, n_total))

    def offspring_from_scan_sharded(self, n, totals, rank, world, seed, frame, n_total, first):
        t = [int(v) for v in totals.tolist()]
        base, total = sum(t[:rank]), sum(t)
        u = oracle.comb_offset(seed, frame, total)
        first[:n] = torch.from_numpy(oracle.offspring_offsets(self._cdf, base, total, u, n_total))

    def offspring_offsets(self, cdf, n, d_base, d_total, seed, frame, n_total, first):
        base = int(d_base[0]) if d_base is not None else 0
        total = int(d_total[0])
        u = oracle.comb_offset(seed, frame, total)
        first[:n] = torch.from_numpy(oracle.offspring_offsets(_np(cdf).view(np.uint64), base, total, u, n_total))

    def ancestors(self, first_all, n_total, slot0, nslots, anc):
        anc[:nslots] = torch.from_numpy(oracle.ancestors(_np(first_all)[:n_total], slot0, nslots))

    def ancestors_from_scan(self, n, seed, frame, anc):
        first = torch.zeros(n, dtype=torch.int32)
        self.offspring_from_scan(n, None, None, seed, frame, n, first)
        self.ancestors(first, n, 0, n, anc)

    def ancestors_sharded(self, first_all, n_total, n_local, rank, world, src, plan, pose_idx=None):
        """Independent restatement with numpy set operations: rank s sends rank r the DISTINCT ancestors (its own
        particles) of r's slots, in particle order; r stages them in rank order behind its n_local particles."""
        fa = _np(first_all)[:n_total]
        n = n_local
        g_all = oracle.ancestors(fa, 0, n_total).astype(np.int64)            # ancestor of every slot of every rank
        owner_all = g_all // n
        bounds = [int(fa[q * n]) for q in range(world)] + [n_total]
        out_plan = np.zeros(1 + 3 * world, np.int32)
        out_plan[0] = int(any(bounds[q] != q * n for q in range(1, world)))
        if self.exch_cap is not None:   # bit 1: could ANY rank's staging area overflow (bound: its slots with a remote ancestor)
            for q in range(world):
                lo, hi = max(bounds[q], q * n), min(bounds[q + 1], (q + 1) * n)
                if n - max(hi - lo, 0) > self.exch_cap:
                    out_plan[0] |= 2
        mine = g_all[rank * n:(rank + 1) * n]
        out = np.empty(n, np.int64)
        off = 0
        for s_ in range(world):
            sel = owner_all[rank * n:(rank + 1) * n] == s_
            if s_ == rank:
                out[sel] = mine[sel] - rank * n
                continue
            uniq, inv = np.unique(mine[sel], return_inverse=True)
            out[sel] = n + off + inv
            out_plan[1 + world + s_] = len(uniq)
            off += len(uniq)
        self._send_rows = []
        for d in range(world):
            slots = g_all[d * n:(d + 1) * n]
            uniq = np.unique(slots[(owner_all[d * n:(d + 1) * n] == rank)]) if d != rank else np.empty(0, np.int64)
            self._send_rows.append((uniq - rank * n).astype(np.int64))
            out_plan[1 + d] = len(uniq)
            # send_base: 1-based rank of the first sent particle among this rank's particles that have offspring
            if len(uniq):
                cnt = np.diff(np.append(fa[rank * n:(rank + 1) * n], fa[(rank + 1) * n] if (rank + 1) * n < n_total else n_total))
                out_plan[1 + 2 * world + d] = int(np.count_nonzero(cnt[: int(uniq[0] - rank * n) + 1] > 0))
        src[:n] = torch.from_numpy(out.astype(np.int32))
        plan[:] = torch.from_numpy(out_plan)
        if pose_idx is not None:   # position of the ancestor's pose in the all-gathered [rank][x|y|theta][n] array
            pose_idx[:n] = torch.from_numpy(((mine // n) * 3 * n + mine % n).astype(np.int32))

    def read_plan(self, d_plan, world):
        return d_plan.tolist()

    def migrate_pack(self, n_local, rank, world, plan, pose, pose_ld, mp, row_stride, plane_stride, nlandmarks, out):
        rec, off = 3 + 5 * nlandmarks, 0
        for d in range(world):
            loc = torch.from_numpy(self._send_rows[d])
            c = len(loc)
            assert c == plan[1 + d]
            if not c:
                continue
            blk = out[rec * off: rec * (off + c)].view(c, rec)
            blk[:, :3] = pose[:, loc].T
            if nlandmarks:
                blk[:, 3:] = mp[loc][:, :, :nlandmarks].reshape(c, 5 * nlandmarks)
            off += c

    def migrate_unpack(self, inp, world, recv_cnt, n_local, pose, pose_ld, mp, row_stride, plane_stride, nlandmarks):
        rec, tot = 3 + 5 * nlandmarks, int(sum(recv_cnt))
        if not tot:
            return
        blk = inp[: rec * tot].view(tot, rec)
        pose[:, n_local: n_local + tot] = blk[:, :3].T
        if nlandmarks:
            mp[n_local: n_local + tot, :, :nlandmarks] = blk[:, 3:].view(tot, 5, nlandmarks)

    def gather_f32(self, src, idx, n, dst):
        dst[:n] = src[idx[:n].long()]

    def gather_map(self, m_in, m_out, in_row_stride, out_row_stride, in_plane_stride, out_plane_stride, nlandmarks, idx, n):
        m_out[:n, :, :nlandmarks] = m_in[idx[:n].long()][:, :, :nlandmarks]
