"""CPU model of the LDS plane images behind the transposing-read weight gradients (csrc/ndq_mlp.h: tr_slot, tr_store,
hbar_wgrad_tr, hbar_wgrad_tr64).  The lane map of ds_read_b64_tr_b16 is the one measured on MI355X by
scripts/ubench_tr16.hip (profiles/archive/r03/r03zz_tr16_lane_map.log): result j of lane l = 16-bit element (l & 3) of the 8 bytes
addressed by lane (l & ~15) + 4 j + ((l & 15) >> 2).  With it, the writer's and the reader's address formulas are replayed
on a byte-addressed model of LDS and every lane must end up with exactly the (point, unit) operand elements the MFMA
contraction needs -- and the bank properties the layout was chosen for must hold."""
import itertools


def tr_slot(row, chunk):                 # csrc/ndq_mlp.h tr_slot: offset in floats inside one 1 KB image
    return chunk * 64 + ((row ^ (4 * chunk)) * 4)


def tr_read(lds, addr_bytes):
    """ds_read_b64_tr_b16 of one wave: addr_bytes[l] per lane -> 4 sixteen-bit elements per lane."""
    out = []
    for l in range(64):
        vals = []
        for j in range(4):
            src = (l & ~15) + 4 * j + ((l & 15) >> 2)
            vals.append(lds[addr_bytes[src] + 2 * (l & 3)])
        out.append(vals)
    return out


def write_plane(lds, base_bytes, tag):
    """tr_store of one plane: lane (p = point, q) holds units 4q..4q+3 of block 0 then of block 1 -> 16 bytes at tr_slot(p, q).
    Elements are stored as tuples (tag, point, unit) at even byte addresses."""
    for lane in range(64):
        p, q = lane & 15, lane >> 4
        a = base_bytes + 4 * tr_slot(p, q)
        for e in range(8):
            unit = 16 * (e >> 2) + 4 * q + (e & 3)
            lds[a + 2 * e] = (tag, p, unit)


def test_h32_operands_of_the_16x16x32_weight_gradient_mfma():
    lds = {}
    PL = 256 * 4                                              # one plane image in bytes
    # a round of two streams: images [stream][plane]
    for s, k in itertools.product(range(2), range(3)):
        write_plane(lds, (s * 3 + k) * PL, ("z", s, k))
    for k, b in itertools.product(range(3), range(2)):       # plane k, unit block b
        addr0, addr1 = [], []
        for lane in range(64):
            kg, i = lane >> 4, lane & 15
            strm = kg >> 1
            r0 = tr_slot(4 * (kg & 1) + (i >> 2), i & 3)
            r1 = tr_slot(4 * (kg & 1) + 8 + (i >> 2), i & 3)
            base = (strm * 3 + k) * PL
            addr0.append(base + 4 * (r0 + 2 * b))
            addr1.append(base + 4 * (r1 + 2 * b))
        lo, hi = tr_read(lds, addr0), tr_read(lds, addr1)
        for lane in range(64):
            kg, i = lane >> 4, lane & 15
            for h, vals in enumerate((lo[lane], hi[lane])):
                for j, v in enumerate(vals):
                    # contraction slot 8 kg + 4 h + j <-> stream kg >> 1, point 4 (kg & 1) + 8 h + j; row of the operand = unit 16 b + i
                    assert v == (("z", kg >> 1, k), 4 * (kg & 1) + 8 * h + j, 16 * b + i), (lane, h, j, v)


def test_h64_operands_of_the_32x32x16_weight_gradient_mfma():
    lds = {}
    PL = 512 * 4
    for k, c in itertools.product(range(3), range(2)):       # plane k, chunk c = units 32 c .. 32 c + 31: an H = 32 image each
        for lane in range(64):
            p, q = lane & 15, lane >> 4
            a = k * PL + c * 1024 + 4 * tr_slot(p, q)
            for e in range(8):
                lds[a + 2 * e] = (k, p, 32 * c + 16 * (e >> 2) + 4 * q + (e & 3))
    for k, blk in itertools.product(range(3), range(2)):
        addr0, addr1 = [], []
        for lane in range(64):
            i16, kg, bsel = lane & 15, lane >> 5, (lane >> 4) & 1
            r0 = tr_slot(8 * kg + (i16 >> 2), i16 & 3) + 2 * bsel
            r1 = tr_slot(8 * kg + 4 + (i16 >> 2), i16 & 3) + 2 * bsel
            addr0.append(k * PL + blk * 1024 + 4 * r0)
            addr1.append(k * PL + blk * 1024 + 4 * r1)
        lo, hi = tr_read(lds, addr0), tr_read(lds, addr1)
        for lane in range(64):
            i32, kg = lane & 31, lane >> 5
            for h, vals in enumerate((lo[lane], hi[lane])):
                for j, v in enumerate(vals):
                    # 32x32x16 operand: lane (i = l & 31, kg = l >> 5) holds unit 32 blk + i at the points 8 kg + 4 h + j
                    assert v == (k, 8 * kg + 4 * h + j, 32 * blk + i32), (lane, h, j, v)


def test_bank_properties_of_the_plane_image():
    # ds_write_b128 is serviced 8 lanes at a time: the 8 lanes of a group must cover 128 contiguous bytes (32 banks)
    for q in range(4):
        for g in range(2):
            addrs = sorted(4 * tr_slot(p, q) for p in range(8 * g, 8 * g + 8))
            assert addrs == list(range(addrs[0], addrs[0] + 128, 16)) and addrs[0] % 128 == 0
    # the 4 rows x 4 chunks a 16-lane group of the transposing read gathers lie in 16 different 16-byte slots (mod 256 B)
    for row0 in (0, 4, 8, 12):
        slots = {(4 * tr_slot(row0 + (i >> 2), i & 3) // 16) % 16 for i in range(16)}
        assert len(slots) == 16
    # the image is a permutation of its 64 sixteen-byte slots
    assert sorted(tr_slot(p, q) for p in range(16) for q in range(4)) == list(range(0, 256, 4))
