"""LDS bank-conflict model after the banking rules of /opt/skills/guides/MI355X_MICROARCH.md (LDS section), for the two layouts it found
conflicts in (round 5) and the ones that replaced them.

Rules used: a wave64 LDS access is served in fixed lane groups, one LDS cycle per group when conflict-free; N distinct addresses on one bank
within a group cost N cycles.  ds_write_b16 / b32: two 32-lane groups, bank = (byte address / 4) mod 32.  ds_read2_b64: two accesses, each four
contiguous 16-lane groups, same 32 banks.  ds_read_b128: four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59},
{36-43,48-51,60-63} on 16 slots of 16 bytes ((byte address / 16) mod 16).  ds_read_b64_tr_b16: two 32-lane groups, bank = (byte address / 4)
mod 64 (the guide warns of further conflict classes; the kernel's layout was measured, not only modelled).

  python tools/lds_bank_model.py      prints the tables DESIGN.md section 4 quotes
tests/test_host_cpu.py imports the functions to pin the layouts the kernels use."""

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def _worst(addr_by_bank):
    return max((len(v) for v in addr_by_bank.values()), default=0)


# ---- attn_kernel (k_attn.hip), the TRANSPOSED V image of rounds 1-5: sV[d][key], pitch VLD halfs -----------------------------------------
def attn_vt_write_cycles(D, VLD, rot=lambda ch, key: 0):
    """LDS cycles per workgroup and key tile of the eight 2-byte stores per staged chunk (lane -> (key, 8-channel chunk), chunk fastest)."""
    CH = D // 8
    nslot = (64 * CH + 255) // 256
    tot = 0
    for wave in range(4):
        for i in range(nslot):
            for e in range(8):
                for g in range(2):
                    banks = {}
                    for lane in range(32 * g, 32 * g + 32):
                        idx = wave * 64 + lane + i * 256
                        if idx >= 64 * CH:
                            continue
                        key, ch = idx // CH, idx % CH
                        a = 2 * ((ch * 8 + (e + rot(ch, key)) % 8) * VLD + key)
                        banks.setdefault((a // 4) % 32, set()).add(a // 4)
                    tot += _worst(banks)
    return tot


def attn_vt_read_cycles(D, VLD):
    """LDS cycles per wave and key tile of the ds_read2_b64 fragment reads of the transposed image."""
    DVF = (D + 31) // 32
    tot = 0
    for kk in range(4):
        for f in range(DVF):
            for half in range(2):
                for g in range(4):
                    banks = {}
                    for lane in range(16 * g, 16 * g + 16):
                        lq, hh = lane & 31, lane >> 5
                        a = 2 * ((f * 32 + lq) * VLD + kk * 16 + 4 * hh + 8 * half)
                        for dw in range(2):
                            banks.setdefault((a // 4 + dw) % 32, set()).add(a // 4 + dw)
                    tot += _worst(banks)
    return tot


# ---- attn_kernel, the ROW-MAJOR V image read through ds_read_b64_tr_b16: sV[key][d], pitch VLD halfs ------------------------------------
def attn_tr_read_cycles(D, VLD):
    """LDS cycles per wave and key tile of the transposing reads (two per fragment and 16-key step), basic bank rule only."""
    DVF = (D + 31) // 32
    tot = 0
    for kk in range(4):
        for f in range(DVF):
            for second in range(2):
                for g in range(2):
                    banks = {}
                    for lane in range(32 * g, 32 * g + 32):
                        hh, i = lane >> 5, lane & 15
                        a = 2 * ((kk * 16 + 8 * second + 4 * hh + (i >> 2)) * VLD + f * 32 + ((lane >> 4) & 1) * 16 + 4 * (i & 3))
                        for dw in range(2):
                            banks.setdefault((a // 4 + dw) % 64, set()).add(a // 4 + dw)
                    tot += _worst(banks)
    return tot


# ---- conv3x_kernel (k_conv3x.hip): ds_read_b128 of the activation fragments from the XOR-swizzled halo tile -------------------------------
def conv3x_halo_read_cycles(key_of_row, IW=16):
    """Average LDS cycles per 16-lane group (1.0 = conflict-free) over every wave, fragment, tap and k-step; key_of_row(halo row) -> XOR key."""
    HWD = IW + 2
    tot = n = 0
    for wave in range(4):
        for p in range(2):
            for tap in range(9):
                for kk in range(4):
                    for g in B128_GROUPS:
                        slots = {}
                        for lane in g:
                            pl, h = lane & 31, lane >> 5
                            if IW == 16:
                                centre = (4 * wave + 2 * p + (pl >> 4) + 1) * HWD + (pl & 15) + 1
                            else:
                                centre = wave * HWD * HWD + (4 * p + (pl >> 3) + 1) * HWD + (pl & 7) + 1
                            row = centre + (tap // 3 - 1) * HWD + (tap % 3 - 1)
                            a = row * 128 + (((2 * kk + h) ^ (key_of_row(row) & 7)) << 4)
                            slots.setdefault((a // 16) % 16, set()).add(a)
                        tot += _worst(slots)
                        n += 1
    return tot / n


def conv3x_key_linear(IW):       # rounds 1-5: from the linear halo row
    return lambda r: r >> 1


def conv3x_key_position(IW):     # halo_key<IW> of k_conv3x.hip: from the pixel's position inside its (IW + 2)^2 halo block
    HWD = IW + 2
    if IW == 16:
        return lambda r: (r % HWD) >> 1
    return lambda r: (((r % HWD) >> 1) + 4 * (((r % (HWD * HWD)) // HWD) & 1))


def main():
    print("attn_kernel, transposed V image: LDS cycles per workgroup and key tile (stores + 4 waves x fragment reads)")
    for D in (40, 80):
        for VLD in (64, 68, 72, 76):
            w, r = attn_vt_write_cycles(D, VLD), attn_vt_read_cycles(D, VLD)
            print(f"  d = {D:3d}  pitch {VLD}: stores {w:5d}  reads/wave {r:5d}  total {w + 4 * r:5d}")
        print(f"  d = {D:3d}  pitch 76 + element order rotated by chunk >> 1: stores {attn_vt_write_cycles(D, 76, lambda ch, key: ch >> 1)}")
    print("attn_kernel, row-major V image + ds_read_b64_tr_b16: reads per wave and key tile (32 = conflict-free at d = 40, 48 at d = 80)")
    for D, pitches in ((40, (64, 72, 96, 160)), (80, (96, 104, 160)), (160, (160, 168))):
        for VLD in pitches:
            print(f"  d = {D:3d}  pitch {VLD}: {attn_tr_read_cycles(D, VLD)}")
    print("conv3x_kernel, activation fragment reads: average LDS cycles per 16-lane group (1.0 = conflict-free)")
    for IW in (16, 8):
        print(f"  {IW} x {IW} tiles: key from the linear row {conv3x_halo_read_cycles(conv3x_key_linear(IW), IW):.3f}, "
              f"from the position in the halo block {conv3x_halo_read_cycles(conv3x_key_position(IW), IW):.3f}")


if __name__ == "__main__":
    main()
