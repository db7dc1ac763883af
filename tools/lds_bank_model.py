"""LDS bank-conflict model of attn_kernel's V^T image (k_attn.hip) after the banking rules of /opt/skills/guides/MI355X_MICROARCH.md (LDS section):
ds_write_b16 / b32: two 32-lane groups, bank = (byte address / 4) mod 32; ds_read2_b64: two accesses of four contiguous 16-lane groups, same banks;
an N-way conflict costs N LDS cycles for its group.  Prints the cycles per workgroup and key tile of the transposing 2-byte stores and of the
fragment reads for candidate row pitches (VLD, in halfs) and for rotated element orders.  python tools/lds_bank_model.py"""
import itertools
def wr_cycles(D, VLD, mapping="ch_fast"):
    CH = D // 8
    nslot = (64*CH + 255)//256
    tot = 0
    for wave in range(4):
      for i in range(nslot):
        for e in range(8):
          for g in range(2):  # 32-lane groups
            banks = {}
            for lane in range(32*g, 32*g+32):
                idx = wave*64 + lane + i*256
                if idx >= 64*CH: continue
                if mapping == "ch_fast": key, ch = idx // CH, idx % CH
                else: key, ch = idx % 64, idx // 64
                a = 2*((ch*8+e)*VLD + key)
                banks.setdefault((a//4) % 32, set()).add(a//4)
            tot += max([len(v) for v in banks.values()], default=0)
    return tot  # LDS cycles per WG per tile for the b16 writes
def rd_cycles(D, VLD):
    DVF = (D+31)//32
    tot = 0
    for kk in range(4):
      for f in range(DVF):
        for half in range(2):  # two accesses of read2_b64 (vrow, vrow+8)
          for g in range(4):   # 16-lane contiguous groups
            banks = {}
            for lane in range(16*g, 16*g+16):
                lq, hh = lane & 31, lane >> 5
                a = 2*((f*32+lq)*VLD + kk*16 + 4*hh + 8*half)
                for dw in range(2):
                    banks.setdefault((a//4 + dw) % 32, set()).add(a//4 + dw)
            tot += max(len(v) for v in banks.values())
    return tot  # per wave per tile
for D in (40, 80):
    print("D", D)
    for VLD in range(64, 97, 4):
        print("  VLD", VLD, "write cycles/WG-tile", wr_cycles(D, VLD), " read cycles/wave-tile", rd_cycles(D, VLD), " total/WG-tile", wr_cycles(D,VLD) + 4*rd_cycles(D,VLD))

def wr_cycles_rot(D, VLD, rot):
    CH = D // 8
    nslot = (64*CH + 255)//256
    tot = 0
    for wave in range(4):
      for i in range(nslot):
        for e in range(8):
          for g in range(2):
            banks = {}
            for lane in range(32*g, 32*g+32):
                idx = wave*64 + lane + i*256
                if idx >= 64*CH: continue
                key, ch = idx // CH, idx % CH
                ee = (e + rot(ch, key)) % 8
                a = 2*((ch*8+ee)*VLD + key)
                banks.setdefault((a//4) % 32, set()).add(a//4)
            tot += max([len(v) for v in banks.values()], default=0)
    return tot
print("rotation variants (write cycles per WG-tile):")
for D in (40, 80):
    for VLD in (68, 76):
        for name, rot in (("none", lambda ch, key: 0), ("ch", lambda ch, key: ch), ("2ch", lambda ch, key: 2*ch), ("ch>>1", lambda ch, key: ch >> 1), ("key", lambda ch, key: key), ("key>>1", lambda ch, key: key >> 1), ("ch+key>>1", lambda ch, key: ch + (key >> 1))):
            print("  D", D, "VLD", VLD, name, wr_cycles_rot(D, VLD, rot))


# ---- conv3x (k_conv3x.hip): ds_read_b128 of the activation fragments from the swizzled halo tile, 16-lane groups on 16 slots of 16 B
GROUPS = [list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
          list(range(32,36))+list(range(44,48))+list(range(52,60)), list(range(36,44))+list(range(48,52))+list(range(60,64))]
def cycles(swz, IW=16):
    HWD = IW + 2
    tot = 0; n = 0
    for wave in range(4):
      for p in range(2):
        for tap in range(9):
          for kk in range(4):
            for g in GROUPS:
                slots = {}
                for lane in g:
                    pl, h = lane & 31, lane >> 5
                    if IW == 16: centre = (4*wave + 2*p + (pl >> 4) + 1)*HWD + (pl & 15) + 1
                    else: centre = wave*HWD*HWD + (4*p + (pl >> 3) + 1)*HWD + (pl & 7) + 1
                    row = centre + (tap//3 - 1)*HWD + (tap % 3 - 1)
                    chunk = 2*kk + h
                    a = row*128 + ((chunk ^ swz(row)) << 4)
                    slots.setdefault((a // 16) % 16, set()).add(a)
                tot += max(len(v) for v in slots.values()); n += 1
    return tot / n
cands = {
 "(row>>1)&7 [current]": lambda r: (r >> 1) & 7,
 "row&7": lambda r: r & 7,
 "(row>>1 ^ row>>4)&7": lambda r: ((r >> 1) ^ (r >> 4)) & 7,
 "((row>>1)+(row//18))&7": lambda r: ((r >> 1) + r // 18) & 7,
 "((row%18)>>1 + 3*(row//18))&7": lambda r: (((r % 18) >> 1) + 3*(r // 18)) & 7,
 "(row*5>>1)&7": lambda r: ((r*5) >> 1) & 7,
}
for IW in (16, 8):
    print("IW", IW)
    for k, f in cands.items():
        print(f"  {k:34s} avg cycles per 16-lane group: {cycles(f, IW):.3f}")

print("--- (y, x) based swizzles")
def mk(IW, fx):
    HWD = IW + 2
    return lambda r: fx((r % (HWD*HWD)) // HWD if IW == 8 else r // HWD, r % HWD) & 7
for IW in (16, 8):
    print("IW", IW)
    for name, fx in {"x>>1": lambda y, x: x >> 1, "x>>1 + 4(y&1)": lambda y, x: (x >> 1) + 4*(y & 1), "x>>1 + 2y": lambda y, x: (x >> 1) + 2*y,
                     "x>>1 + 4y": lambda y, x: (x >> 1) + 4*y, "x>>1 + 2(y&3)": lambda y, x: (x >> 1) + 2*(y & 3), "x>>1 + 6y": lambda y, x: (x >> 1) + 6*y,
                     "(x>>1) ^ 4(y&1)": lambda y, x: (x >> 1) ^ (4*(y & 1)), "x>>1 + 3y": lambda y, x: (x >> 1) + 3*y, "x>>1 + 5y": lambda y, x: (x >> 1) + 5*y}.items():
        print(f"  {name:20s} {cycles(mk(IW, fx), IW):.3f}")
