#!/usr/bin/env python3
"""Per-layer achieved TFLOP/s of the conv3x3_mfma launches in a rocprofv3 rocpd DB of bench.py (one pipeline step =
64 conv launches with 6B samples (MISO1) followed by 64 with 2B samples (MISO3)).  usage: conv_layer_report.py db [B] [T]"""
import sqlite3
import sys

EN = (24, 32, 32, 32, 32, 64, 128); DE = (128, 64, 32, 32, 32, 32, 24)
FE = [127, 63, 31, 15, 7, 3, 1]


def layers(in_ch, out_ch):
    L = []
    def dense(tag, c0, g1, g2, F):
        for i in range(5):
            L.append((f"{tag}.c{i+1}", c0 + i * g1, g1 if i < 4 else g2, F, F))
    en = [in_ch] + list(EN)
    for b in range(7):
        Fin = 129 if b == 0 else FE[b - 1]
        L.append((f"enc{b}.conv", en[b], en[b + 1], FE[b], FE[b]))
        if b < 5:
            dense(f"enc{b}.db", EN[b], EN[b], EN[b], FE[b])
    de = list(DE) + [out_ch]
    for i in range(7):
        Fi = FE[6 - i]
        if i >= 2:
            dense(f"dec{i}.db", 2 * DE[i], DE[i], 2 * DE[i], Fi)
        L.append((f"dec{i}.deconv", 2 * DE[i], de[i + 1], Fi, Fi))     # transposed: MACs counted on input positions
    return L


def main():
    db = sqlite3.connect(sys.argv[1])
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 1001
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    conv = []
    for n, s, e in rows:
        if "conv3x3_" not in n:
            continue
        # f32w runs a layer with a 16-channel group (dec6.db.c5: 48 = 32 + 16) as TWO launches: the 32-row body, then the 16-row
        # instantiation conv3x3_wino_f32<0, true> -- one layer in this report
        if "conv3x3_wino_f32<0, true>" in n.replace("(bool)1", "true") and conv:
            conv[-1] = (conv[-1][0], conv[-1][1] + (e - s))
        else:
            conv.append((n, e - s))
    assert len(conv) >= 128, len(conv)      # (+ the two launches of the f32w self-check at the first commit)
    last = conv[-128:]
    l1, l3 = layers(12, 4), layers(16, 2)
    print(f"{'layer':14s} {'Cin':>4s} {'Cout':>4s} {'F':>4s} {'ms(MISO1 x%d)' % (6*B):>14s} {'TF/s':>7s} {'ms(MISO3 x%d)' % (2*B):>14s} {'TF/s':>7s}")
    tot = [0, 0, 0, 0]
    for k in range(64):
        nm, cin, cout, F, _ = l1[k]
        cin3 = l3[k][1]; cout3 = l3[k][2]
        f1 = 2.0 * cin * cout * 9 * F * T * 6 * B
        f3 = 2.0 * cin3 * cout3 * 9 * F * T * 2 * B
        d1, d3 = last[k][1] / 1e6, last[64 + k][1] / 1e6
        tot[0] += d1; tot[1] += f1; tot[2] += d3; tot[3] += f3
        print(f"{nm:14s} {cin:4d} {cout:4d} {F:4d} {d1:14.3f} {f1 / d1 / 1e9:7.1f} {d3:14.3f} {f3 / d3 / 1e9:7.1f}")
    print(f"total MISO1 {tot[0]:.1f} ms {tot[1]/tot[0]/1e9:.1f} TF/s ; MISO3 {tot[2]:.1f} ms {tot[3]/tot[2]/1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
