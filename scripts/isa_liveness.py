"""VGPR liveness over a kernel's gfx950 assembly (hipcc -S): where is the register-pressure peak and what is live there?

usage: isa_liveness.py file.s kernel-name-substring [n_top]
Conservative def/use model (first operand = destination except stores / compares; MFMA, fmac and DPP-with-old read their destination)."""
import re
import sys


def regs(tok):
    tok = tok.strip().split()[0] if tok.strip() else ""
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def defuse(line):
    parts = line.split(None, 1)
    if len(parts) < 2:
        return set(), set()
    op = parts[0]
    ops = [o for o in re.split(r",\s*(?![^\[]*\])", parts[1].split(";")[0])]
    d, u = set(), set()
    nodst = op.startswith(("ds_write", "global_store", "scratch_store", "buffer_store", "global_atomic", "s_", "v_cmp", "global_load_lds", "v_readfirstlane", "v_readlane", "ds_bpermute_never"))
    for k, o in enumerate(ops):
        rs = regs(o)
        if k == 0 and not nodst:
            d |= rs
            if op.startswith(("v_mfma", "v_fmac", "v_pk_fmac", "v_mac", "v_dot")) or "dpp" in line or "sdwa" in line or op.startswith(("v_cndmask", "v_writelane")):
                u |= rs if (op.startswith(("v_fmac", "v_pk_fmac", "v_mac", "v_writelane")) or "dpp" in line) else set()
        else:
            u |= rs
    return d, u


def main():
    text = open(sys.argv[1]).read()
    name = [n for n in re.findall(r"^(_Z\w+):", text, re.M) if sys.argv[2] in n][0]
    body = text[text.index(name + ":"):]
    body = body[:body.index(".end_amdhsa_kernel")]
    lines = []
    for ln in body.split("\n"):
        ln = ln.split(";")[0].strip()
        if not ln:
            continue
        if re.match(r"^\.LBB\w+:$", ln) or not ln.startswith("."):
            lines.append(ln)
    label_at = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}
    n = len(lines)
    succ = [[] for _ in range(n)]
    for i, ln in enumerate(lines):
        op = ln.split()[0]
        if op == "s_endpgm":
            continue
        if op == "s_branch":
            succ[i].append(label_at[ln.split()[1]])
            continue
        if op.startswith("s_cbranch"):
            succ[i].append(label_at[ln.split()[1]])
        if i + 1 < n:
            succ[i].append(i + 1)
    du = [defuse(ln) if not ln.endswith(":") else (set(), set()) for ln in lines]
    live_in = [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            out = set()
            for s in succ[i]:
                out |= live_in[s]
            new = (out - du[i][0]) | du[i][1]
            if new != live_in[i]:
                live_in[i] = new
                changed = True
    order = sorted(range(n), key=lambda i: -len(live_in[i]))
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    shown = []
    for i in order:
        if all(abs(i - j) > 40 for j in shown):
            shown.append(i)
            mf = sum(1 for ln in lines[:i] if ln.startswith("v_mfma"))
            print(f"--- {len(live_in[i])} live VGPRs before instruction {i} (after {mf} MFMAs): {lines[i]}")
            # last definition of each live register
            groups = {}
            for r in sorted(live_in[i]):
                dline = next((j for j in range(i - 1, -1, -1) if r in du[j][0]), None)
                groups.setdefault(dline, []).append(r)
            for dline, rs in sorted(groups.items(), key=lambda kv: (kv[0] is None, kv[0])):
                print(f"   v{rs[0]}..v{rs[-1]} ({len(rs)})  <- {dline}: {lines[dline] if dline is not None else '?'}")
            if len(shown) >= top:
                break


if __name__ == "__main__":
    main()
