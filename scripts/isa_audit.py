"""ISA audit of a -save-temps assembly file: per kernel (1) the sequence of waits / barriers / DMA / loads / MFMA groups,
(2) for every asm-issued (hidden) global load, that no instruction touches its destination registers before an asm
s_waitcnt statement (the counted wait that names them) has executed.  usage: isa_audit.py file.s [kernel-substring] [-v]"""
import re
import sys


def regs_of(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit(body, verbose):
    """Walk one kernel.  `queue` models the in-order vmcnt queue (loads, stores, atomics and LDS-DMA of either origin);
    an entry carries the VGPRs an asm-issued load will write.  s_waitcnt vmcnt(N) retires all but the N youngest."""
    in_asm = False
    queue = []  # list of sets of destination vgprs (empty set for DMA / stores / compiler loads)
    bad = []
    out = []
    cnt = dict(mfma=0, ds=0, gl=0, dma=0, hid=0)

    def flush():
        if any(cnt.values()):
            out.append("[" + " ".join(f"{v}{k}" for k, v in cnt.items() if v) + "]")
            for k in cnt:
                cnt[k] = 0

    def retire(n):
        while len(queue) > n:
            queue.pop(0)

    for n, line in enumerate(body):
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        t = t.split(";")[0].strip()
        op = t.split()[0]
        toks = re.findall(r"v\[\d+:\d+\]|v\d+", t)
        m = re.search(r"vmcnt\((\d+)\)", t)
        if op == "s_waitcnt":
            if m:
                retire(int(m.group(1)))
                flush()
                out.append(("A:" if in_asm else "C:") + t.replace("s_waitcnt ", ""))
            continue
        if op == "s_barrier":
            flush()
            out.append("BAR" if in_asm else "cBAR")
            continue
        if in_asm and op.startswith("global_load_lds"):
            queue.append(set())
            cnt["dma"] += 1
            continue
        if in_asm and op.startswith("global_load"):
            queue.append(regs_of(toks[0]))
            cnt["hid"] += 1
            continue
        # compiler instruction: must not touch registers of asm loads that are still in flight
        touched = set()
        for tk in toks:
            touched |= regs_of(tk)
        pending = set().union(*queue) if queue else set()
        hit = touched & pending
        if hit:
            bad.append((n, t, sorted(hit)))
        if op.startswith("v_mfma"):
            cnt["mfma"] += 1
        elif op.startswith("ds_"):
            cnt["ds"] += 1
        elif op.startswith("global_load_lds"):
            cnt["dma"] += 1
            queue.append(set())
        elif op.startswith("global_load") or op.startswith("scratch_load"):
            cnt["gl"] += 1
            queue.append(set())
        elif op.startswith("global_store") or op.startswith("global_atomic") or op.startswith("scratch_store"):
            queue.append(set())
            flush()
            out.append(op)
    flush()
    if verbose:
        print(" ".join(out))
    return bad


def main():
    s = open(sys.argv[1]).read()
    verbose = "-v" in sys.argv
    args = [x for x in sys.argv[2:] if x != "-v"]
    sel = args[0] if args else ""
    for m in re.finditer(r"^(_Z\w+):.*\n", s, re.M):
        name = m.group(1)
        if sel not in name or "kernel" not in name:
            continue
        end = s.index(".Lfunc_end", m.end())
        body = s[m.end():end].split("\n")
        print("==", name)
        bad = audit(body, verbose)
        print("   hidden-load register hazards:", len(bad))
        for n, t, hit in bad[:10]:
            print("     line", n, t[:90], hit)


main()
