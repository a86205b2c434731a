"""ISA audit of a -save-temps assembly file: per kernel (1) the sequence of waits / barriers / DMA / loads / MFMA groups,
(2) for every asm-issued (hidden) global load, that no instruction touches its destination registers before an asm
s_waitcnt statement (the counted wait that names them) has executed.  usage: isa_audit.py file.s [kernel-substring]"""
import re
import sys


def regs_of(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit(body, verbose):
    in_asm = False
    pending = {}  # vgpr -> line of hidden load
    bad = []
    out = []
    cnt = dict(mfma=0, ds=0, gl=0, dma=0, hid=0)

    def flush():
        if any(cnt.values()):
            out.append("[" + " ".join(f"{v}{k}" for k, v in cnt.items() if v) + "]")
            for k in cnt:
                cnt[k] = 0

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
        op = t.split()[0]
        toks = re.findall(r"v\[\d+:\d+\]|v\d+", t)
        if in_asm and op.startswith("global_load_lds"):
            cnt["dma"] += 1
            continue
        if in_asm and op.startswith("global_load"):
            for r in regs_of(toks[0]):
                pending[r] = n
            cnt["hid"] += 1
            continue
        if in_asm and op == "s_waitcnt":
            flush()
            out.append("A:" + t.replace("s_waitcnt ", ""))
            # conservative: an asm wait releases every pending register older than it that it names via "+v"; the
            # operands are not visible in the text, so release all pending loads issued before this wait only if the
            # wait count allows: we cannot know -> release all that a later reader could legally see (checked by count
            # logic in the kernel design).  Here: release everything (order check only).
            pending_snapshot = dict(pending)
            pending.clear()
            continue
        if in_asm and op == "s_barrier":
            flush()
            out.append("BAR")
            continue
        # compiler instruction: must not touch pending hidden-load registers
        touched = set()
        for tk in toks:
            touched |= regs_of(tk)
        hit = touched & set(pending)
        if hit:
            bad.append((n, t, sorted(hit)))
        if op.startswith("v_mfma"):
            cnt["mfma"] += 1
        elif op.startswith("ds_"):
            cnt["ds"] += 1
        elif op.startswith("global_load_lds"):
            cnt["dma"] += 1
        elif op.startswith("global_load"):
            cnt["gl"] += 1
        elif op == "s_waitcnt" and "vmcnt" in t:
            flush()
            out.append("C:" + t.replace("s_waitcnt ", ""))
        elif op == "s_barrier":
            flush()
            out.append("cBAR")
        elif op.startswith("global_store") or op.startswith("global_atomic") or op.startswith("scratch_"):
            flush()
            out.append(op)
    flush()
    if verbose:
        print(" ".join(out))
    return bad


def main():
    s = open(sys.argv[1]).read()
    sel = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"^(_Z\w+):.*\n", s, re.M):
        name = m.group(1)
        if sel not in name or "kernel" not in name:
            continue
        end = s.index(".Lfunc_end", m.end())
        body = s[m.end():end].split("\n")
        print("==", name)
        bad = audit(body, True)
        print("   hidden-load register hazards:", len(bad))
        for n, t, hit in bad[:10]:
            print("     line", n, t[:90], hit)


main()
