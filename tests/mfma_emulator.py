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
    def __init__(self, flat, feat=256, tau=4, bf16=False):
        m = packing.forward_maps(feat, tau)
        self.m, self.tau, self.auxs, self.bf16 = m, tau, m["auxs"], bf16
        flat = np.asarray(flat, np.float32)
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

    def _stage(self, frags, aux, n_tiles, act):
        out = []
        for _ in range(n_tiles):
            acc = act(self._tile(frags + aux))
            out += [self._q(acc[:, :8]), self._q(acc[:, 8:])]
        return out

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
        cur = []
        for s in range(16):
            sig = 16 * s + 8 * h[:, None] + np.arange(8)[None, :]
            w = self.l0[sig]  # [64, 8, 4]
            pre = w[..., 0] * xyz[p, 0:1] + w[..., 1] * xyz[p, 1:2] + w[..., 2] * xyz[p, 2:3] + w[..., 3]
            cur.append(self._q(sin_rev(pre)))
        for _ in range(7):
            cur = self._stage(cur, aux, 8, sin_rev)
        feats = self._stage(cur, aux, 8, lambda v: v)
        sig_acc = self._tile(cur + aux)
        softplus = lambda v: np.where(v > 20, v, np.log1p(np.exp(np.minimum(v, 20))))  # noqa: E731
        sigmoid = lambda v: 1 / (1 + np.exp(-v))  # noqa: E731
        sigma = softplus(sig_acc[:32, 0])
        c = np.zeros((64, 16))
        rgbh = self._stage(feats, aux, 4, sin_rev)
        c += self._tile(rgbh)
        s1 = self._stage(feats, aux, 4, sin_rev)
        s2 = self._stage(s1, aux, 4, sin_rev)
        s3 = self._stage(s2, aux, 4, sin_rev)
        c += self._tile(s3)
        e1 = self._stage(feats, aux, 4, sin_rev)
        acc = c + self._tile(e1 + aux)
        assert self.cur == self.stream.shape[0], (self.cur, self.stream.shape)
        albedo = sigmoid(acc[:32, 0:3]) * 1.002 - 0.001
        return albedo, sigma, sigmoid(acc[:32, 3]), softplus(acc[32:, 0])
