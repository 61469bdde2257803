"""Static check of a gfx950 assembly listing for the hazard class that inline-asm LDS reads open up.

The register-resident chain (dynhor_amd/csrc/chain_t.hip) issues its LDS reads as inline asm (a read the compiler can see makes
it drain every LDS-DMA in flight) and waits for them itself.  The compiler does not know such a read's destination registers
are still in flight, so nothing stops it from (a) scheduling a use of them above the wait, or (b) handing them, when nobody
reads the value, to something else that is live when the data lands.  Both happened during bring-up (wrong activations from
one layer on, values off by 1e-2).  This scanner walks the listing: every `ds_read*` marks its destination registers pending,
`s_waitcnt ... lgkmcnt(0)` clears them; any instruction that reads or overwrites a pending register in between is reported.
Counted waits (round 5: the training chain leaves its patch writes in flight behind `lgkmcnt(N)`): the LDS operations of a wave
complete in order, so `lgkmcnt(N)` completes all but the N youngest `ds_*` operations -- unless a scalar memory operation (which
shares the counter and returns out of order) is outstanding, in which case only `lgkmcnt(0)` is trusted.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only chain_t.hip -o chain_t.s && python scripts/isa_inflight_check.py chain_t.s
"""
import re
import sys

_STORE = ("ds_write", "global_store", "buffer_store", "scratch_store", "flat_store")


def _regs(text):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        out += list(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else [int(m.group(3))]
    return out


def scan(lines):
    """-> (number of ds_read instructions seen, list of (line number, text, register, 'read'|'write', line of the ds_read))"""
    pending, found, n_reads = {}, [], 0
    queue = []            # outstanding LDS operations, oldest first: the destination registers of a read, () for anything else
    smem = False          # a scalar memory operation is outstanding: counted lgkmcnt waits say nothing about the LDS ones
    for i, raw in enumerate(lines, 1):
        t = raw.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        if t.startswith("s_waitcnt"):
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                if n == 0:
                    pending, queue, smem = {}, [], False
                elif not smem and len(queue) > n:
                    for regs in queue[:len(queue) - n]:
                        for r in regs:
                            pending.pop(r, None)
                    queue = queue[len(queue) - n:]
            continue
        m = re.match(r"(\S+)\s+(.*)", t)
        if not m:
            continue
        op, parts = m.group(1), [a.strip() for a in m.group(2).split(",")]
        if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_store", "s_buffer_store", "s_dcache")):
            smem = True
        if op.startswith("ds_read"):
            n_reads += 1
            for r in _regs(",".join(parts[1:])):
                if r in pending:
                    found.append((i, t, r, "read", pending[r]))
            dst = _regs(parts[0])
            for r in dst:
                pending[r] = i
            queue.append(tuple(dst))
            continue
        if op.startswith("ds_"):
            queue.append(())
        is_store = op.startswith(_STORE)
        for r in _regs(",".join(parts if is_store else parts[1:])):
            if r in pending:
                found.append((i, t, r, "read", pending[r]))
        if not is_store:
            for r in _regs(parts[0]):
                if r in pending:
                    found.append((i, t, r, "write", pending[r]))
    return n_reads, found


if __name__ == "__main__":
    n, found = scan(open(sys.argv[1]).read().split("\n"))
    for f in found[:40]:
        print("line %d: %s  -- %s of v%d, in flight since line %d" % (f[0], f[1], f[3], f[2], f[4]))
    print("%d ds_read instructions, %d hazards" % (n, len(found)))
    sys.exit(1 if found else 0)
