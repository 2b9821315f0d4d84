"""CPU model of the NickNet kernel's data flow (csrc/nicknet.hip), evaluated from the PACKED weight blob.

The kernel keeps every layer output as a "tile": unit 16 b + 4 i + q in register [b][i] of the lanes of group
q, window on the lane's low four bits.  A v_mfma_f32_16x16x4_f32 takes, for k-step ks, the value lane group q
holds for that k-step (k = q) and leaves row 4 q' + i' of its 16 x 16 result in register i' of lane group q'.
This module restates exactly that algebra with numpy (float64) on the blob vad_pack_nicknet_weights writes --
table layout [group of 4 k-steps][out block][lane][4], the GRU's recurrent weights as float16 split operands of
v_mfma_f32_16x16x32_f16 with their power-of-two scales, bias tiles [block][q][i], the pre-scaled activations --
so the packer and the lane algebra can be checked against the oracle without a GPU
(tests/test_nicknet_lane_model.py).
"""
import numpy as np

T, F, XK = 30, 24, 6
SPLIT_BLOCK = 3 * 64 * 4                          # floats of one out block of a float16 split table
STATE_SCALE = 2.0 ** 14                           # the split state is 2^14 h
LOG2E = 1.4426950408889634


def n_blocks(nu):
    return (nu + 15) // 16


def block_ksteps(nu, b):
    return (max(0, min(16, nu - 16 * b)) + 3) // 4


def tile_ksteps(nu):
    return block_ksteps(nu, 0) + block_ksteps(nu, 1)


def groups(ks):
    return (ks + 3) // 4


def last_width(ks):
    """Floats per lane of a table's last group of k-steps: one or two k-steps are stored at that width."""
    return {1: 1, 2: 2}.get(ks % 4, 4)


def table_floats(ks, nout):
    return (groups(ks) - 1) * nout * 256 + nout * 64 * last_width(ks)


class Dims:
    def __init__(self, large):
        self.large = large
        self.chan = [F, 32, 28, 25, 18] if large else [F, 20, 18, 16]
        self.nc = len(self.chan) - 1
        self.layers = 2 if large else 1
        self.hidden = 29 if large else 21
        self.ne = 2 if large else 1
        self.embed_in = [self.hidden, 32, 18] if large else [self.hidden, 18]
        self.embed_out = [32, 18, 2] if large else [18, 2]

    def conv_ks(self, l):
        return XK if l == 0 else tile_ksteps(self.chan[l])

    def gru_in(self, g):
        return self.chan[self.nc] if g == 0 else self.hidden


def layout(d):
    L, at = {}, 0
    for l in range(d.nc):
        tap = table_floats(d.conv_ks(l), 2 * n_blocks(d.chan[l + 1]))
        L['conv', l], L['conv_tap', l] = at, tap
        at += 3 * tap
    gout = 3 * n_blocks(d.hidden)
    for g in range(d.layers):
        L['gru_ih', g] = at                       # layers above the first: a float16 split table like gru_hh
        at += table_floats(tile_ksteps(d.gru_in(g)), gout) if g == 0 else gout * SPLIT_BLOCK
        L['gru_hh', g] = at                       # float16 split table: [out block][hi, mid, lo][64 lanes] x 8 float16
        at += gout * SPLIT_BLOCK
    for l in range(d.nc):
        L['conv_bias', l] = at
        at += 64
    for g in range(d.layers):
        L['gru_bias', g] = at
        at += 128
    for g in range(d.layers):
        L['gru_scale', g] = at                    # 2^-(14 + Sw), 2^Sw of gru_hh, then of a split gru_ih
        at += 4
    L['lds'] = at
    for s in range(d.ne + 1):
        L['emb', s] = at
        L['emb_tab', s] = table_floats(tile_ksteps(d.embed_in[s]), n_blocks(d.embed_out[s]))
        at += (d.layers if s == 0 else 1) * L['emb_tab', s]
        L['emb_bias', s] = at
        at += 32
    L['total'] = at
    return L


# A tile is an array [2 blocks][4 i][4 q][windows]: register [b][i] of lane group q.

def kvals(tile, nu):
    """The B-operand values per k-step: list of arrays [4 q][windows]."""
    out = []
    for b in range(2):
        for i in range(block_ksteps(nu, b)):
            out.append(tile[b, i])
    return out


def mm(acc, blob, off, nout, kv, at=lambda o: o):
    """acc[at(o)] (each [4 i][4 q][windows]) += table(out block o) . kv -- one MFMA per (group, k-step, out block)."""
    ks_total = len(kv)
    full, lw = groups(ks_total) - 1, last_width(ks_total)
    tab = np.zeros((groups(ks_total), nout, 64, 4))
    tab[:full] = blob[off: off + full * nout * 256].reshape(full, nout, 64, 4)
    tab[full, :, :, :lw] = blob[off + full * nout * 256: off + table_floats(ks_total, nout)].reshape(nout, 64, lw)
    for g in range(groups(ks_total)):
        for j in range(4):
            ks = 4 * g + j
            if ks >= ks_total:
                continue
            for o in range(nout):
                a = tab[g, o, :, j].reshape(4, 16)                   # [k = lane group][row m]
                res = np.einsum('km,kw->mw', a, kv[ks])              # (16 rows, windows)
                acc[at(o)] += res.reshape(4, 4, -1).transpose(1, 0, 2)   # row 4 q' + i' -> [i'][q']
    return acc


def mm_split(acc, blob, off, nout, tile, unscale, at=lambda o: o):
    """acc[at(o)] += unscale * (split table(out block o) . 2^14 tile): the recurrent product as the kernel forms it
    with v_mfma_f32_16x16x32_f16 -- lane 16 qa + m of the A operand holds row m and k-slots 8 qa + j, which are the
    units lane group qa keeps in register [j >> 2][j & 3]; the three float16 terms of a weight are summed here (their
    six kept partial products differ from the full product by less than 2^-35 of it)."""
    raw = blob[off: off + nout * SPLIT_BLOCK].view(np.uint16).view(np.float16).reshape(nout, 3, 64, 8).astype(np.float64)
    a = raw.sum(axis=1).reshape(nout, 4, 16, 8)                           # [o][qa][row m][j]
    for o in range(nout):
        res = np.zeros((16, tile.shape[-1]))
        for qa in range(4):
            for j in range(8):
                res += a[o, qa, :, j][:, None] * (STATE_SCALE * tile[j >> 2, j & 3, qa])[None, :]
        acc[at(o)] += unscale * res.reshape(4, 4, -1).transpose(1, 0, 2)   # row 4 q' + i' -> [i'][q']
    return acc


def bias_tile(blob, off, windows):
    """[2 blocks][4 i][4 q][windows] from the stored [block][q][i]."""
    t = blob[off: off + 32].reshape(2, 4, 4).astype(np.float64).transpose(0, 2, 1)
    return np.repeat(t[:, :, :, None], windows, axis=3).copy()


def sigmoid_pre(a):
    return 1.0 / (1.0 + np.exp2(a))


def tanh_pre(a):
    return 1.0 - 2.0 / (1.0 + np.exp2(a))


def tile_to_units(tile, nu):
    """[2][4 i][4 q][w] -> (w, nu): unit 16 b + 4 i + q."""
    w = tile.shape[-1]
    out = np.zeros((w, nu))
    for u in range(nu):
        out[:, u] = tile[u // 16, (u % 16) // 4, u % 4]
    return out


def forward(blob, x, large):
    """x (B,30,24) -> dict(trunk (B,C,30), seq (B,30,H), hn (L,B,H), logits (B,2), p (B,2)), float64."""
    d = Dims(large)
    L = layout(d)
    assert L['total'] == blob.size, (L['total'], blob.size)
    B = x.shape[0]
    x = x.astype(np.float64)

    def x_kv(pos):                                    # lane group q supplies feature 6 q + s for k-step s
        return [np.stack([x[:, pos, XK * q + s] for q in range(4)]) for s in range(XK)]

    def conv(l, prev_kv):                             # prev_kv: dict position -> k-step values of the layer input
        outs = {}
        nb = n_blocks(d.chan[l + 1])
        for pos in range(T):
            acc = np.zeros((2 * nb, 4, 4, B))
            for vg in range(2):
                acc[vg * nb: vg * nb + nb] = bias_tile(blob, L['conv_bias', l] + 32 * vg, B)[:nb]
            for tap in range(3):
                src = pos + tap - 1
                if 0 <= src < T:
                    mm(acc, blob, L['conv', l] + tap * L['conv_tap', l], 2 * nb, prev_kv[src])
            tile = np.zeros((2, 4, 4, B))
            tile[:nb] = tanh_pre(acc[:nb]) * sigmoid_pre(acc[nb:])
            outs[pos] = tile
        return outs

    kv = {pos: x_kv(pos) for pos in range(T)}
    for l in range(d.nc):
        tiles = conv(l, kv)
        kv = {pos: kvals(tiles[pos], d.chan[l + 1]) for pos in range(T)}
    C, H = d.chan[d.nc], d.hidden
    trunk = np.stack([tile_to_units(tiles[pos], C) for pos in range(T)], axis=2)      # (B, C, 30)

    nb = n_blocks(H)
    h = [np.zeros((2, 4, 4, B)) for _ in range(d.layers)]
    seq = np.zeros((B, T, H))
    for t in range(T):
        xin = kv[t]
        for g in range(d.layers):
            acc = np.zeros((4 * nb, 4, 4, B))
            for a in range(4):
                acc[a * nb: a * nb + nb] = bias_tile(blob, L['gru_bias', g] + 32 * a, B)[:nb]
            if g == 0:
                mm(acc, blob, L['gru_ih', g], 3 * nb, xin)
            else:
                mm_split(acc, blob, L['gru_ih', g], 3 * nb, h[g - 1], float(blob[L['gru_scale', g] + 2]))
            if t > 0:
                mm_split(acc, blob, L['gru_hh', g], 3 * nb, h[g], float(blob[L['gru_scale', g]]),
                         at=lambda o: o if o < 2 * nb else o + nb)
            r, z = sigmoid_pre(acc[:nb]), sigmoid_pre(acc[nb:2 * nb])
            n = tanh_pre(r * acc[3 * nb:] + acc[2 * nb:3 * nb])
            h[g][:nb] = z * (h[g][:nb] - n) + n
            xin = kvals(h[g], H)
        seq[:, t] = tile_to_units(h[-1], H)
    hn = np.stack([tile_to_units(hg, H) for hg in h])
    logits = embed_tiles(blob, L, d, h, B)
    e = np.exp(logits - logits.max(1, keepdims=True))
    return {'trunk': trunk, 'seq': seq, 'hn': hn, 'logits': logits, 'p': e / e.sum(1, keepdims=True)}


def embed_tiles(blob, L, d, h, B):
    e = None
    for s in range(d.ne + 1):
        nbo = n_blocks(d.embed_out[s])
        acc = bias_tile(blob, L['emb_bias', s], B)[:nbo].copy()
        if s == 0:
            for l in range(d.layers):
                mm(acc, blob, L['emb', 0] + l * L['emb_tab', 0], nbo, kvals(h[l], d.hidden))
        else:
            mm(acc, blob, L['emb', s], nbo, kvals(e, d.embed_in[s]))
        if s < d.ne:
            e = np.zeros((2, 4, 4, B))
            e[:nbo] = np.maximum(acc, 0.0)
    return tile_to_units(np.concatenate([acc, np.zeros((1, 4, 4, B))]), 2)


def embed(blob, v, large):
    """v (B, L*H) -> logits (B,2)."""
    d = Dims(large)
    L = layout(d)
    B = v.shape[0]
    h = []
    for l in range(d.layers):
        tile = np.zeros((2, 4, 4, B))
        for u in range(d.hidden):
            tile[u // 16, (u % 16) // 4, u % 4] = v[:, l * d.hidden + u]
        h.append(tile)
    return embed_tiles(blob, L, d, h, B)
