"""Lane-accurate numpy emulation of the fused forward kernel's dataflow (csrc/mlp_fwd.hip).

Used by the CPU tests to validate the weight-stream gather maps, the slot permutation, the aux k-step and the
head wiring WITHOUT a GPU: it walks the packed stream piece by piece exactly as the kernel does, with the
v_mfma_f32_32x32x16_bf16 operand/accumulator lane maps (A: row = lane&31, k = 8*(lane>>5)+j; B: col = lane&31,
same k; C/D: col = lane&31, row = (g&3) + 8*(g>>2) + 4*(lane>>5)).  fp64 arithmetic, optional bf16 rounding of
the MFMA operands.
"""
import numpy as np

from satnerf_amd import packing

LANE = np.arange(64)
ROW_OF = (np.arange(16)[None, :] & 3) + 8 * (np.arange(16)[None, :] >> 2) + 4 * (LANE[:, None] >> 5)  # [lane, g]


def bf16_round(x):
    x = np.asarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).astype(np.float64)


class Emulator:
    def __init__(self, flat, feat=256, tau=4, bf16=False, l0_split=False):
        self.l0_split = l0_split  # fc_net.0 as the generated core computes it: two k-steps of three-way bf16 splits (gen/fwd_core.py l0_terms)
        m = packing.forward_maps(feat, tau)
        self.m, self.tau, self.auxs, self.bf16, self.m_feat = m, tau, m["auxs"], bf16, feat
        flat = np.asarray(flat, np.float32)
        self.flat = flat
        idx = m["idx"]
        vals = np.where(idx >= 0, flat[np.maximum(idx, 0)] * m["scale"], np.float32(0)).astype(np.float32)
        self.stream = (bf16_round(vals) if bf16 else vals.astype(np.float64)).reshape(-1, 2, 32, 8)  # piece, h, row, j
        self.l0 = (flat[m["l0_idx"]] * m["l0_scale"]).astype(np.float32).astype(np.float64).reshape(feat, 4)

    def _q(self, x):
        return bf16_round(x) if self.bf16 else np.asarray(x, np.float64)

    def _tile(self, frags):
        """one output tile: consume len(frags) pieces; frags are B fragments [64 lanes, 8]."""
        c = np.zeros((32, 32))
        for b in frags:
            a = self.stream[self.cur]
            self.cur += 1
            c += np.einsum("hrj,hcj->rc", a, b.reshape(2, 32, 8))
        return c[ROW_OF, (LANE & 31)[:, None]]  # acc[lane, g]

    def _stage(self, frags, aux, n_tiles, act, tag=None):
        out, pres = [], []
        for _ in range(n_tiles):
            raw = self._tile(frags + aux)
            acc = act(raw)
            out += [self._q(acc[:, :8]), self._q(acc[:, 8:])]
            pres += [raw[:, :8], raw[:, 8:]]
        if tag is not None:
            self.saved["pre"][tag] = pres  # pre-activations in revolutions, fragment layout
        return out

    @staticmethod
    def _split3(v):
        v = np.asarray(v, np.float32)
        h = bf16_round(v).astype(np.float32)
        m = bf16_round((v - h).astype(np.float32)).astype(np.float32)
        lo = bf16_round((v - h - m).astype(np.float32)).astype(np.float32)
        return h.astype(np.float64), m.astype(np.float64), lo.astype(np.float64)

    def _l0_split_pre(self, w, x):
        """w [64, 8, 4] table rows, x [64, 3] -> pre-activation: the small cross terms w_m x_m + w_l x_h accumulated first (one MFMA,
        rounded to fp32), then w_h (x_h + x_m + x_l) + w_m x_h and the three bias parts (second MFMA)"""
        small, big = np.zeros(w.shape[:2]), np.zeros(w.shape[:2])
        for c in range(3):
            wh, wm, wl = self._split3(w[..., c])
            xh, xm, xl = (v[:, None] for v in self._split3(x[:, c]))
            small += wm * xm + wl * xh
            big += wh * xh + wh * xm + wm * xh + wh * xl
        bh, bm, bl = self._split3(w[..., 3])
        big += bh + bm + bl
        return (small.astype(np.float32).astype(np.float64) + big).astype(np.float32).astype(np.float64)

    def forward_tile(self, xyz, sun, t):
        """xyz, sun (32,3), t (32,tau) -> albedo (32,3), sigma, sun_v, beta (32,)"""
        self.cur = 0
        p, h = LANE & 31, LANE >> 5
        sin_rev = lambda v: np.sin(2 * np.pi * v)  # noqa: E731
        aux = []
        for a in range(self.auxs):
            v = np.zeros((64, 8))
            for j in range(8):
                q = 16 * a + 8 * h + j
                ti = q - 8
                ok = (q >= 8) & (ti < self.tau)
                v[:, j] = np.where(ok, t[p, np.clip(ti, 0, self.tau - 1)], 0.0)
            if a == 0:
                h0 = h == 0
                first = np.concatenate([sun[p], np.ones((64, 1)), xyz[p], np.zeros((64, 1))], 1)
                v[h0] = first[h0]
            aux.append(self._q(v))
        sv = self.saved = {"aux": aux, "a": [], "pre": {}}
        cur = []
        pre0 = []
        nks, nt, nth = self.m_feat // 16, self.m_feat // 32, self.m_feat // 64  # k-steps / output tiles of a trunk layer, tiles of a head layer
        for s in range(nks):
            sig = 16 * s + 8 * h[:, None] + np.arange(8)[None, :]
            w = self.l0[sig]  # [64, 8, 4]
            pre = w[..., 0] * xyz[p, 0:1] + w[..., 1] * xyz[p, 1:2] + w[..., 2] * xyz[p, 2:3] + w[..., 3]
            if self.l0_split:
                pre = self._l0_split_pre(w, xyz[p])
            cur.append(self._q(sin_rev(pre)))
            pre0.append(pre)
        sv["a"].append(cur)
        sv["pre"]["a0"] = pre0
        for l in range(1, 8):
            cur = self._stage(cur, aux, nt, sin_rev, tag=f"a{l}")
            sv["a"].append(cur)
        feats = self._stage(cur, aux, nt, lambda v: v)
        sv["feats"] = feats
        sig_acc = self._tile(cur + aux)
        softplus = lambda v: np.where(v > 20, v, np.log1p(np.exp(np.minimum(v, 20))))  # noqa: E731
        sigmoid = lambda v: 1 / (1 + np.exp(-v))  # noqa: E731
        sigma = softplus(sig_acc[:32, 0])
        c = np.zeros((64, 16))
        rgbh = self._stage(feats, aux, nth, sin_rev, tag="rgbh")
        c += self._tile(rgbh)
        s1 = self._stage(feats, aux, nth, sin_rev, tag="s1")
        s2 = self._stage(s1, aux, nth, sin_rev, tag="s2")
        s3 = self._stage(s2, aux, nth, sin_rev, tag="s3")
        c += self._tile(s3)
        e1 = self._stage(feats, aux, nth, sin_rev, tag="e1")
        sv.update(rgbh=rgbh, s1=s1, s2=s2, s3=s3, e1=e1)
        acc = c + self._tile(e1 + aux)
        assert self.cur == self.stream.shape[0], (self.cur, self.stream.shape)
        albedo = sigmoid(acc[:32, 0:3]) * 1.002 - 0.001
        sv["out"] = (albedo, sigma, sigmoid(acc[:32, 3]), softplus(acc[32:, 0]))
        return sv["out"]

    # ------------------------------------------------------------------------------------------- backward
    def backward_tile(self, g_albedo, g_sigma, g_sun, g_beta):
        """Emulates csrc/mlp_bwd.inc (dX chain over the transposed stream) + csrc/wgrad.hip (block GEMMs) + the gradient
        gather; returns the flat fp64 gradient vector (sky parameters = 0).  Call forward_tile first."""
        bm = packing.backward_maps(256, self.tau)
        flat = self.flat
        idx = bm["idx"]
        vals = np.where(idx >= 0, flat[np.maximum(idx, 0)].astype(np.float64) * bm["scale"], 0.0)
        bstream = (bf16_round(vals) if self.bf16 else vals).reshape(-1, 2, 32, 8)
        cur = [0]

        def tile(frags):
            c = np.zeros((32, 32))
            for b in frags:
                c += np.einsum("hrj,hcj->rc", bstream[cur[0]], b.reshape(2, 32, 8))
                cur[0] += 1
            return c[ROW_OF, (LANE & 31)[:, None]]

        sv = self.saved
        albedo, sigma, sun_v, beta = sv["out"]
        p, h = LANE & 31, LANE >> 5
        cosr = lambda pre: np.cos(2 * np.pi * pre)  # noqa: E731
        dhead = np.zeros((64, 8))
        sg = (albedo + 0.001) / 1.002
        for c in range(3):
            dhead[:32, c] = g_albedo[:, c] * 1.002 * sg[:, c] * (1 - sg[:, c])
        dhead[:32, 3] = g_sun * sun_v * (1 - sun_v)
        dhead[32:, 0] = g_beta * (1 - np.exp(-beta))
        dsig = np.zeros((64, 8))
        dsig[:32, 0] = g_sigma * (1 - np.exp(-sigma))
        dhead, dsig = self._q(dhead), self._q(dsig)

        def bstage(frags, n_tiles, pres):
            out = []
            for t in range(n_tiles):
                acc = tile(frags)
                if pres is not None:
                    acc = acc * np.concatenate([cosr(pres[2 * t]), cosr(pres[2 * t + 1])], 1)
                out += [self._q(acc[:, :8]), self._q(acc[:, 8:])]
            return out

        pre = sv["pre"]
        d_rgbh = bstage([dhead], 4, pre["rgbh"])
        d_s3 = bstage([dhead], 4, pre["s3"])
        d_e1 = bstage([dhead], 4, pre["e1"])
        d_s2 = bstage(d_s3, 4, pre["s2"])
        d_s1 = bstage(d_s2, 4, pre["s1"])
        d_feats = bstage(d_rgbh + d_s1 + d_e1, 8, None)
        dt_acc = tile(d_e1)  # rows = t index
        d_pre = [None] * 8
        d_pre[7] = bstage(d_feats + [dsig], 8, pre["a7"])
        for l in range(7, 0, -1):
            d_pre[l - 1] = bstage(d_pre[l], 8, pre[f"a{l - 1}"])
        assert cur[0] == bstream.shape[0]
        # workspaces in fragment order (mlp_layout.h)
        acts = list(sv["aux"]) + [f for l in range(8) for f in sv["a"][l]] + sv["feats"] + sv["rgbh"] + sv["s1"] + sv["e1"] + sv["s2"] + sv["s3"]
        dpre = [f for l in range(8) for f in d_pre[l]] + d_feats + [dsig] + d_rgbh + d_s1 + d_e1 + d_s2 + d_s3 + [dhead]
        slotmat = lambda fr: np.concatenate([f.reshape(2, 32, 8).transpose(0, 2, 1).reshape(16, 32) for f in fr], 0)  # noqa: E731
        partial = np.zeros((bm["blocks"].shape[0], packing.WG_BLOCK_FLOATS))
        aux_m = slotmat(acts[0:self.auxs])
        for b, (rows, cols) in enumerate(zip(bm["block_rows"], bm["block_cols"])):
            nr, nc = len(rows), len(cols)
            r = slotmat([dpre[f] for f in rows])
            main, aux = partial[b, :256 * 256].reshape(256, 256), partial[b, 256 * 256:].reshape(256, 32)
            if nc > 0:
                main[:16 * nr, :16 * nc] = r @ slotmat([acts[f] for f in cols]).T
            aux[:16 * nr, :16 * self.auxs] = r @ aux_m.T
        pf = partial.reshape(-1)
        grad = np.where(bm["gidx"] >= 0, pf[np.maximum(bm["gidx"], 0)] * bm["gscale"], 0.0)
        # rows of the d-t tile: lane (p,h) reg g holds row ROW_OF
        d_t = np.zeros((32, self.tau))
        for lane in range(64):
            for g in range(16):
                r = ROW_OF[lane, g]
                if r < self.tau:
                    d_t[lane & 31, r] = dt_acc[lane, g]
        return grad, d_t
