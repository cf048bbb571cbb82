"""numpy emulation of the register-resident MFMA dataflow (adanerf_amd/csrc/layout.hpp): consumes
the library's packed A fragments exactly as the kernels do, so the weight packer and the slot
permutation are checked on CPU against the oracle's plain matmuls."""
import ctypes as C

import numpy as np


def pack_weights(lib, model_dir, net, precision):
    wb, bf, nl = C.c_size_t(0), C.c_size_t(0), C.c_int32(0)
    f = lib.adanerf_host_pack_weights
    f.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p,
                  C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_int32)]
    f.restype = C.c_int
    rc = f(model_dir.encode(), net, precision, None, C.byref(wb), None, C.byref(bf), None, C.byref(nl))
    assert rc == 0, lib.adanerf_last_error(None)
    w = np.empty(wb.value, dtype=np.uint8)
    b = np.empty(bf.value, dtype=np.float32)
    lay = np.empty((nl.value, 4), dtype=np.int32)
    rc = f(model_dir.encode(), net, precision, w.ctypes.data, C.byref(wb), b.ctypes.data, C.byref(bf),
           lay.ctypes.data, C.byref(nl))
    assert rc == 0, lib.adanerf_last_error(None)
    return w, b, lay


def _decode(raw, precision):
    if precision == 2:
        return raw.view(np.float32)
    u = raw.view(np.uint16)
    if precision == 0:   # bf16
        return (u.astype(np.uint32) << 16).view(np.float32)
    return u.view(np.float16).astype(np.float32)


def quantize(x, precision):
    """round-to-nearest-even to the MFMA operand type (what ET::pack does on device)"""
    x = np.asarray(x, dtype=np.float32)
    if precision == 2:
        return x
    if precision == 1:
        return x.astype(np.float16).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def pe_slots(F):
    return ((3 * F + 2) + 7) & ~7


def pe_eval(x, F):
    """x [n,3] -> slots [2, pe_slots(F), n] (lane-half, slot, sample)"""
    n = x.shape[0]
    out = np.zeros((2, pe_slots(F), n), dtype=np.float32)
    for q in range(3 * F):
        b, c = divmod(q, 3)
        a = (x[:, c] * np.float32(2.0 ** b)).astype(np.float32)
        out[0, q] = np.sin(a)
        out[1, q] = np.cos(a)
    out[0, 3 * F] = x[:, 0]
    out[1, 3 * F] = x[:, 2]
    out[0, 3 * F + 1] = x[:, 1]
    return out


def f16(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


class PackedNet:
    def __init__(self, w, b, lay, precision):
        self.w, self.b, self.lay, self.precision = w, b, lay, precision
        self.G = 4 if precision == 2 else 8
        # a bf16 shading net is packed SCALED (pack.cpp scale_layer): one more record {alpha exponent, rgb exponent, 0, -1}; every ReLU
        # layer's outputs must then stay <= 1 -- the kernels' clamped conversion would cut them off (checked in layer())
        self.scaled = lay.shape[0] > 0 and int(lay[-1][3]) == -1
        self.out_exp = (int(lay[-1][0]), int(lay[-1][1])) if self.scaled else (0, 0)
        if self.scaled:
            self.lay = lay[:-1]
        self.max_relu_out = 0.0

    def unscale(self, rgb_rows, alpha):
        """the kernels' store_raw: outputs back in the network's own scale (exact powers of two)"""
        return [np.ldexp(r, self.out_exp[1]).astype(np.float32) for r in rgb_rows], np.ldexp(alpha, self.out_exp[0]).astype(np.float32)

    def layer_split(self, l, act, relu):
        """precision 3: (hi, lo') fragment pairs; W.x = Whi.xhi + (Whi.xlo' + Wlo'.xhi) / 2048"""
        w_off, b_off, QS, MT = [int(v) for v in self.lay[l]]
        steps = QS // 8
        frag = self.w[w_off * 16:(w_off + MT * steps * 2 * 64) * 16].view(np.float16).astype(np.float32)
        frag = frag.reshape(MT, steps, 2, 64, 8)
        a_hi = f16(act)
        a_lo = f16((act - a_hi) * np.float32(2048.0))
        n = act.shape[2]
        out = np.zeros((2, 16 * MT, n), dtype=np.float32)
        bias = self.b[b_off:b_off + MT * 32].reshape(MT, 2, 16)
        for m in range(MT):
            main = np.zeros((32, n), dtype=np.float32)
            cross = np.zeros((32, n), dtype=np.float32)
            for h in range(2):
                Whi = frag[m, :, 0, h * 32:(h + 1) * 32, :].transpose(1, 0, 2).reshape(32, QS)
                Wlo = frag[m, :, 1, h * 32:(h + 1) * 32, :].transpose(1, 0, 2).reshape(32, QS)
                main += Whi @ a_hi[h]
                cross += Whi @ a_lo[h] + Wlo @ a_hi[h]
            D = main + cross / np.float32(2048.0)
            for hh in range(2):
                for r in range(16):
                    row = (r & 3) + 8 * (r >> 2) + 4 * hh
                    out[hh, 16 * m + r] = D[row] + bias[m, hh, r]
        return np.maximum(out, 0) if relu else out

    def layer(self, l, act, relu):
        """act [2, QS, n] -> next slots [2, 16*MT, n]"""
        if self.precision == 3:
            return self.layer_split(l, act, relu)
        w_off, b_off, QS, MT = [int(v) for v in self.lay[l]]
        G = self.G
        steps = QS // G
        assert act.shape[1] == QS, (act.shape, QS)
        frag = _decode(self.w[w_off * 16:(w_off + MT * steps * 64) * 16], self.precision).reshape(MT, steps, 64, G)
        act_q = quantize(act, self.precision)
        n = act.shape[2]
        out = np.zeros((2, 16 * MT, n), dtype=np.float32)
        bias = self.b[b_off:b_off + MT * 32].reshape(MT, 2, 16)
        for m in range(MT):
            D = np.zeros((32, n), dtype=np.float32)
            for h in range(2):
                A = frag[m, :, h * 32:(h + 1) * 32, :]           # [steps, 32 rows, G]
                A = A.transpose(1, 0, 2).reshape(32, QS)         # row i, slot q = G*s + e
                D += A @ act_q[h]
            for hh in range(2):
                for r in range(16):
                    row = (r & 3) + 8 * (r >> 2) + 4 * hh
                    out[hh, 16 * m + r] = D[row] + bias[m, hh, r]
        if relu:
            out = np.maximum(out, 0)
            if self.scaled:
                self.max_relu_out = max(self.max_relu_out, float(out.max()))
                assert out.max() <= 1.0, "a scaled ReLU layer exceeds 1: the kernels' clamped conversion would cut it off"
        return out


def run_sampling_net(net: PackedNet, dir_unit, p, fp, fd):
    act = np.concatenate([pe_eval(dir_unit, fd), pe_eval(p, fp)], axis=1)
    act = net.layer(0, act, True)
    for l in range(1, 7):
        act = net.layer(l, act, True)
    out = net.layer(7, act, False)          # [2, 64, n]
    n = out.shape[2]
    orc = np.zeros((n, 128), dtype=np.float32)
    for h in range(2):
        for q in range(64):
            feat = 32 * (q >> 4) + 8 * ((q & 15) >> 2) + 4 * h + (q & 3)
            orc[:, feat] = out[h, q]
    return orc


def run_shading_net(net: PackedNet, x, dpe, fp=10, fd=4):
    pts = pe_eval(x, fp)
    dirs = pe_eval(dpe, fd)
    h = net.layer(0, pts, True)
    for l in range(1, 5):
        h = net.layer(l, h, True)
    h = net.layer(5, np.concatenate([pts, h], axis=1), True)
    h = net.layer(6, h, True)
    h = net.layer(7, h, True)
    f = net.layer(8, h, False)              # [2, 144, n]; tile 8 row 0 -> half 0, slot 128
    alpha = f[0, 128]
    v = net.layer(9, np.concatenate([f[:, :128], dirs], axis=1), True)
    rgb = net.layer(10, v, False)           # rows 0..2 -> half 0, slots 0..2
    (r, g, b), alpha = net.unscale([rgb[0, 0], rgb[0, 1], rgb[0, 2]], alpha)
    return np.stack([r, g, b, alpha], axis=1)


# ---- any topology, fp32 fragments (the run-time-shaped kernels of k_generic_f32.hip.hpp) -------------------

def run_sampling_net_generic(net: PackedNet, dir_unit, p, nds, fp, fd, rsi_z=None, rsi_d1=1.0):
    """depth = number of layer records (minus the raySampleInput record); layer 0 optionally extended by the K-major
    block of the A extra points p + nds z_a (encode(x / d1), identity slots scaled back by d1)."""
    lay = net.lay
    has_rsi = rsi_z is not None and len(rsi_z) > 0
    assert net.precision == 2 or (net.precision == 3 and not has_rsi)      # split pairs: sample_mlp16x3_gen_kernel
    depth = lay.shape[0] - (1 if has_rsi else 0)
    act0 = np.concatenate([pe_eval(dir_unit, fd), pe_eval(p, fp)], axis=1)
    if not has_rsi:
        act = net.layer(0, act0, True)
    else:
        pre = net.layer(0, act0, False)                                   # bias + first part, slots [2, 16 MT, n]
        w_off, A, QP, MT = [int(v) for v in lay[depth]]
        frag = net.w[w_off * 16:(w_off + A * (QP // 4) * MT * 64) * 16].view(np.float32).reshape(A, QP // 4, MT, 64, 4)
        n = p.shape[0]
        D = np.zeros((MT, 32, n), dtype=np.float32)
        for a in range(A):
            x = ((p + nds * np.float32(rsi_z[a])) / np.float32(rsi_d1)).astype(np.float32)
            t = pe_eval(x, fp)
            t[0, 3 * fp] *= np.float32(rsi_d1)
            t[1, 3 * fp] *= np.float32(rsi_d1)
            t[0, 3 * fp + 1] *= np.float32(rsi_d1)
            for m in range(MT):
                for h in range(2):
                    Am = frag[a, :, m, h * 32:(h + 1) * 32, :].transpose(1, 0, 2).reshape(32, QP)
                    D[m] += Am @ t[h]
        for m in range(MT):
            for hh in range(2):
                for r in range(16):
                    pre[hh, 16 * m + r] += D[m, (r & 3) + 8 * (r >> 2) + 4 * hh]
        act = np.maximum(pre, 0)
    for l in range(1, depth - 1):
        act = net.layer(l, act, True)
    out = net.layer(depth - 1, act, False)
    n = out.shape[2]
    orc = np.zeros((n, 128), dtype=np.float32)
    for h in range(2):
        for q in range(64):
            orc[:, 32 * (q >> 4) + 8 * ((q & 15) >> 2) + 4 * h + (q & 3)] = out[h, q]
    return orc


def run_shading_net_generic(net: PackedNet, x, dpe, depth, width, skip, fp=10, fd=4):
    assert net.precision in (0, 1, 2) and net.lay.shape[0] == depth + 3      # 16-bit: the same fragment order, 8 slots per k-step
    pts, dirs = pe_eval(x, fp), pe_eval(dpe, fd)
    h = net.layer(0, pts, True)
    for l in range(1, depth):
        cat = (l - 1) in skip if isinstance(skip, (list, tuple)) else l == skip + 1      # one skip index or the list of them
        h = net.layer(l, np.concatenate([pts, h], axis=1) if cat else h, True)
    f = net.layer(depth, h, False)                                # feature (+ alpha tile)
    alpha = f[0, width // 2]
    v = net.layer(depth + 1, np.concatenate([f[:, :width // 2], dirs], axis=1), True)
    rgb = net.layer(depth + 2, v, False)
    (r, g, b), alpha = net.unscale([rgb[0, 0], rgb[0, 1], rgb[0, 2]], alpha)
    return np.stack([r, g, b, alpha], axis=1)
