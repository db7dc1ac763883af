"""ISA check for register-staged kernels: for every innermost loop that issues MFMAs, print the order of vector-memory loads,
`s_waitcnt vmcnt(n)` and MFMA groups.  A `vmcnt(0)` between a loop's prefetch loads and its MFMAs means the prefetch is waited
for right after it was issued (the compiler's wait-count pass does that when a value loaded BEFORE the loop is first used
inside it: round 5, attn_kernel's Q fragments).
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o x.s k_x.hip; python tools/isa_loop_waits.py x.s [name filter]"""
import re
import sys


def main():
    path, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    fn, lines = None, {}
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            fn = m.group(1)
            lines[fn] = []
        elif fn:
            lines[fn].append(ln.rstrip())
            if "s_endpgm" in ln:
                fn = None
    for fn, ls in lines.items():
        if flt not in fn:
            continue
        # innermost loops: the header block is tagged "Inner Loop Header", its other blocks "in Loop: Header=<label>"; blocks
        # the loop rotation placed in front of the header are appended behind it (execution order)
        blocks, cur = [], None
        for ln in ls:
            m = re.match(r"^\.L(BB\d+_\d+):(.*)$", ln)
            if m:
                cur = [m.group(1), m.group(2), []]
                blocks.append(cur)
            elif cur is not None:
                cur[2].append(ln)
        for bi, (lab, tag, _) in enumerate(blocks):
            if "Inner Loop Header" not in tag:
                continue
            mine = lambda b: b[0] == lab or re.search(r"Header=" + lab + r"\b", b[1])
            body = [l for b in blocks[bi:] if mine(b) for l in b[2]] + [l for b in blocks[:bi] if mine(b) for l in b[2]]
            if not any("v_mfma" in b for b in body):
                continue
            ev, run = [], 0
            for b in body:
                t = b.strip()
                if t.startswith("v_mfma"):
                    run += 1
                    continue
                if run:
                    ev.append(f"M{run}")
                    run = 0
                if re.match(r"(global_load|buffer_load|flat_load)", t):
                    ev.append("L" + ("lds" if " lds" in t else ""))
                elif t.startswith("s_waitcnt") and "vmcnt" in t:
                    ev.append("W" + re.search(r"vmcnt\((\d+)\)", t).group(1))
                elif t.startswith("s_barrier"):
                    ev.append("|")
            if run:
                ev.append(f"M{run}")
            # compress repeated tokens
            out, prev, cnt = [], None, 0
            for e in ev + [None]:
                if e == prev:
                    cnt += 1
                else:
                    if prev is not None:
                        out.append(prev if cnt == 1 else f"{prev}x{cnt}")
                    prev, cnt = e, 1
            print(f"{fn[:90]} {lab} ({len(body)} lines): {' '.join(out)}")


if __name__ == "__main__":
    main()
