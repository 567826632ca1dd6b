#!/usr/bin/env python3
"""Build-time check of an assumption the fused gradient kernel makes about its own instruction stream.

beta_grads_lds_kernel<KPT, 1, FUSE = true, VL> (csrc/ammsb_beta.hip) waits for the row of edge t with a COUNTED
`s_waitcnt vmcnt((D - 1) * PIECES + ST)`: vmcnt retires in issue order and counts stores too, so the wait must know
how many memory instructions were issued after the request it waits for -- per trip PIECES = KPT / 4 LDS-DMA loads
(the next row) and ST = KPT + 1 stores (the normalised pi row of the edge's partner, one dword per column of the lane,
and phi_sum).  If the compiler ever merged, split, duplicated or dropped one of those stores the count would be off
and a row would be read from LDS before it has landed -- silently.  This script disassembles the gfx950
code object of ammsb_beta.o and requires, for every fused instantiation, that the trip loop (the innermost backward
branch that contains both an LDS-DMA load and a store) holds exactly PIECES `global_load_lds_dwordx4` and executes
exactly ST `global_store_dword` per trip (the pi-row stores come as non-temporal / plain twins behind a wave-uniform
branch: a pair counts once).  Run by csrc/Makefile after ammsb_beta.o is built; exits non-zero on a mismatch.

    python3 tools/check_fused_stores.py mcmc-ammsb-gpu_amd/csrc/ammsb_beta.o
"""
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def device_disassembly(obj):
    with tempfile.TemporaryDirectory() as d:
        local = os.path.join(d, "in.o")
        with open(obj, "rb") as f, open(local, "wb") as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, "--offloading", local], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        dev = [n for n in os.listdir(d) if "amdgcn" in n]
        if not dev:
            raise SystemExit("check_fused_stores: no gfx950 code object found in %s" % obj)
        return subprocess.run([OBJDUMP, "-d", os.path.join(d, dev[0])], capture_output=True, text=True, check=True).stdout


def functions(text):
    cur, out = None, {}
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m:
            out[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def trip_loop(insts):
    """(start, end) index range of the innermost backward-branch loop that holds the function's LAST LDS-DMA load (the
    trip loop's row request; the fused form also fills its ring once, earlier, from a short loop of its own) and a store."""
    addr_to_idx = {a: i for i, (a, _, _) in enumerate(insts)}
    dma = [i for i, (_, o, _) in enumerate(insts) if o.startswith("global_load_lds")]
    last_dma = dma[-1] if dma else -1
    best = None
    for i, (a, op, args) in enumerate(insts):
        if not op.startswith("s_cbranch") and op != "s_branch":
            continue
        m = re.search(r"(-?\d+)\s*$", args)
        if not m:
            continue
        off = int(m.group(1))
        if off >= 32768:
            off -= 65536
        target = a + 4 + 4 * off
        if target > a or target not in addr_to_idx:
            continue
        j = addr_to_idx[target]
        body = insts[j:i + 1]
        if j <= last_dma <= i and any(o.startswith("global_store") for _, o, _ in body):
            if best is None or (i - j) < (best[1] - best[0]):
                best = (j, i)
    return best


def main():
    obj = sys.argv[1]
    funcs = functions(device_disassembly(obj))
    checked, bad = 0, []
    for name, insts in funcs.items():
        m = re.search(r"beta_grads_lds_kernelILi(\d+)ELi1ELb1ELi(\d+)E", name)
        if not m:
            continue
        kpt = int(m.group(1))
        loop = trip_loop(insts)
        if loop is None:
            bad.append((name, "no trip loop found"))
            continue
        body = insts[loop[0]:loop[1] + 1]
        loads = sum(1 for _, o, _ in body if o.startswith("global_load_lds"))
        stores = [(o, a) for _, o, a in body if o.startswith("global_store")]
        wide = [o for o, _ in stores if o != "global_store_dword"]
        # the pi-row stores exist twice, once with the non-temporal hint and once without, behind a wave-uniform branch
        # (BetaArgs.pi_nt): a trip EXECUTES one of each such pair -- the twins are counted once
        plain = {a for _, a in stores if not a.endswith(" nt")}
        twins = sum(1 for _, a in stores if a.endswith(" nt") and a[:-3] in plain)
        executed = len(stores) - twins
        checked += 1
        if loads != kpt // 4 or executed != kpt + 1 or wide:
            bad.append((name, "trip loop has %d LDS-DMA loads (want %d) and executes %d stores per trip (%d in the code, %d "
                              "non-temporal twins; want %d)%s"
                        % (loads, kpt // 4, executed, len(stores), twins, kpt + 1,
                           ", not all single dwords: %s" % wide if wide else "")))
    if not checked:
        raise SystemExit("check_fused_stores: no fused beta_grads_lds_kernel instantiation found in %s" % obj)
    if bad:
        for name, why in bad:
            print("check_fused_stores: %s: %s" % (name, why), file=sys.stderr)
        raise SystemExit(1)
    print("check_fused_stores: %d fused instantiations: LDS-DMA loads and stores per trip as the counted waits assume" % checked)


if __name__ == "__main__":
    main()
