"""LDS bank-conflict model of the split-f16 GEMM's fragment reads (ds_read_b128 lane groups of gfx950, MI355X_MICROARCH.md §LDS)
for a tile row swizzle and a row shift — groundwork for re-using ONE A slab across the five taps of a convolution:
does a fragment read that starts `shift` rows into the slab stay conflict-free?

    python tools/lds_bank_model.py

A tile row is 128 B = 8 slots of 16 B (slots 0-3 the hi plane, 4-7 the lo plane of a 32-column chunk); slot c of tile row r is
stored at slot position c ^ swz(r).  Lane (l31, hi) of a wave reads row l31 + shift, slot plane * 4 + 2 * ks + hi.  A
ds_read_b128 is served in four groups of 16 lanes; a group is conflict-free iff its sixteen 16-B segments fall on sixteen
different bank quads ((address / 16) mod 16)."""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def ways(swz, shift, plane, ks):
    worst = 0
    for g in GROUPS:
        segs = {}
        for lane in g:
            l31, hi = lane & 31, lane >> 5
            row = l31 + shift
            addr = row * 128 + (((plane * 4 + 2 * ks + hi) ^ swz(row)) * 16)
            segs.setdefault((addr // 16) % 16, set()).add(addr)
        worst = max(worst, max(len(v) for v in segs.values()))
    return worst


if __name__ == "__main__":
    cands = {"shipped: c ^ ((r >> 1) & 7)": lambda r: (r >> 1) & 7,
             "c ^ (r & 7)": lambda r: r & 7,
             "c ^ ((r >> 1) & 7) ^ ((r & 1) << 2)": lambda r: ((r >> 1) & 7) ^ ((r & 1) << 2)}
    for name, swz in cands.items():
        res = [max(ways(swz, s, pl, ks) for pl in (0, 1) for ks in (0, 1)) for s in range(5)]
        print(f"{name:40s} worst n-way conflict per row shift 0..4: {res}")
