"""CPU emulation of the DATAFLOW of rdb_bf16_strip_kernel (csrc/rdb_bf16_strip.hip): 16-column strips that sweep down an
image in positions of 12 rows, layer m lagging m-1 rows, every map a circular row buffer of R_l rows with one halo
column each side; neighbours exchange edge columns through per-(layer, position parity) mailboxes.  Checks window sizes,
overwrite hazards, zero masks and the mailbox parity against a plain evaluation of the dense block.  Test tooling only."""
import numpy as np

BH, BW = 12, 16                      # rows per position, columns per strip
R = [18, 17, 16, 15, 14]             # circular rows of x0..x4
CH = [8, 4, 4, 4, 4]                 # channels (scaled down 8x from 64/32 to keep this fast)


def conv3x3(x, w, b):                # x [C,H,W] zero padded, w [O,C,3,3]
    C, H, W = x.shape
    xp = np.zeros((C, H + 2, W + 2)); xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], H, W))
    for dy in range(3):
        for dx in range(3):
            out += np.einsum("oc,chw->ohw", w[:, :, dy, dx], xp[:, dy:dy + H, dx:dx + W])
    return out + b[:, None, None]


def lrelu(v):
    return np.maximum(v, 0.2 * v)


def reference(x0, ws, bs):
    maps = [x0]
    for l in range(4):
        maps.append(lrelu(conv3x3(np.concatenate(maps, 0), ws[l], bs[l])))
    return conv3x3(np.concatenate(maps, 0), ws[4], bs[4]) * 0.2 + x0


class Strip:
    def __init__(self, s, h, w):
        self.s, self.h, self.w, self.xs = s, h, w, s * BW
        self.buf = [np.zeros((CH[l], R[l], BW + 2)) for l in range(5)]        # zeroed once per strip
        self.tagrow = [np.full((R[l],), -10**9) for l in range(5)]              # which image row a slot holds (hazard check)

    def put_row(self, l, y, vals, cols=slice(None)):
        self.buf[l][:, y % R[l], cols] = vals
        if cols == slice(None) or cols == slice(1, BW + 1):
            self.tagrow[l][y % R[l]] = y

    def get_rows(self, l, y0, n):        # rows y0..y0+n-1 with all 18 columns; a slot must hold the row asked for (or never-written zeros for y<0)
        out = np.zeros((CH[l], n, BW + 2))
        for k in range(n):
            y = y0 + k
            t = self.tagrow[l][y % R[l]]
            if y < 0:
                assert t < 0 or t == y, (l, y, t)
            else:
                assert t == y, f"map {l} row {y}: slot holds row {t}"
            out[:, k] = self.buf[l][:, y % R[l]]
        return out


def run(h, w, seed=0):
    rng = np.random.default_rng(seed)
    x0 = rng.standard_normal((CH[0], h, w))
    ws, bs, cin = [], [], CH[0]
    for l in range(5):
        co = CH[l + 1] if l < 4 else CH[0]
        ws.append(rng.standard_normal((co, cin, 3, 3)) * 0.1); bs.append(rng.standard_normal(co) * 0.1)
        cin += co if l < 4 else 0
    ref = reference(x0, ws, bs)
    ns = -(-w // BW)
    npos = -(-(h + 4) // BH)
    strips = [Strip(s, h, w) for s in range(ns)]
    mail = {}                            # (strip, side, layer, parity) -> (pos, [CH, 12]) edge column of x_layer
    out = np.zeros_like(ref)

    def load_x0(st, y_lo, y_hi):
        for y in range(y_lo, y_hi):
            row = np.zeros((CH[0], BW + 2))
            if 0 <= y < h:
                for p in range(BW + 2):
                    x = st.xs - 1 + p
                    if 0 <= x < w:
                        row[:, p] = x0[:, y, x]
            st.put_row(0, y, row)

    for st in strips:
        load_x0(st, 0, BH + 1)
    for pos in range(npos):
        for m in range(1, 6):                       # layer m reads x0..x_{m-1}, writes x_m (m = 5: the output)
            Y0 = BH * pos - (m - 1)
            # halo import of x_{m-1} (m >= 2): rows of layer m-1's block at this position
            for st in strips:
                if m >= 2:
                    l = m - 1
                    Yb = BH * pos - (l - 1)
                    for side, nb, col in ((0, st.s - 1, 0), (1, st.s + 1, BW + 1)):
                        if 0 <= nb < ns:
                            p_, data = mail[(nb, 1 - side, l, pos & 1)]
                            assert p_ == pos, "mailbox overwritten or not yet written"
                            for k in range(BH):
                                st.buf[l][:, (Yb + k) % R[l], col] = data[:, k]
            # compute
            for st in strips:
                ins = np.concatenate([st.get_rows(l, Y0 - 1, BH + 2) for l in range(m)], 0)     # [C, 14, 18]
                acc = np.zeros((ws[m - 1].shape[0], BH, BW))
                for dy in range(3):
                    for dx in range(3):
                        acc += np.einsum("oc,chw->ohw", ws[m - 1][:, :, dy, dx], ins[:, dy:dy + BH, dx:dx + BW])
                acc += bs[m - 1][:, None, None]
                ys = np.arange(Y0, Y0 + BH); xs = np.arange(st.xs, st.xs + BW)
                valid = ((ys >= 0) & (ys < h))[:, None] & (xs < w)[None, :]
                if m < 5:
                    v = np.where(valid[None], lrelu(acc), 0.0)
                    for k in range(BH):
                        st.put_row(m, Y0 + k, v[:, k], slice(1, BW + 1))
                    for side, col in ((0, 0), (1, BW - 1)):
                        mail[(st.s, side, m, pos & 1)] = (pos, v[:, :, col].copy())
                else:
                    res = st.get_rows(0, Y0, BH)[:, :, 1:BW + 1]
                    v = acc * 0.2 + res
                    for k in range(BH):
                        for j in range(BW):
                            if valid[k, j]:
                                out[:, Y0 + k, st.xs + j] = v[:, k, j]
            if m == 5 and pos + 1 < npos:          # x0 rows of the next position: after conv5 has consumed the old ones
                for st in strips:
                    load_x0(st, BH * (pos + 1) + 1, BH * (pos + 1) + 1 + BH)
    err = np.abs(out - ref).max()
    return err


if __name__ == "__main__":
    for (h, w) in [(30, 40), (61, 33), (24, 16), (13, 50), (7, 5)]:
        print(h, w, run(h, w))
