"""ISA check for the limiter's look-ahead poll (rh_limit.hip): an `sc1` poll load issued by inline asm stays in flight across a tile's work; no
instruction may read or overwrite its destination registers before the wait that collects it (s_waitcnt vmcnt(0), or vmcnt(V) with V = C * R / 4
of the instance: the LDS-DMA fetches issued behind it).  The compiler does not know the load is asynchronous: a copy of an asm output (seen:
v_mov_b64 right behind the load, scratch_store of a spilled holder) would read registers the load has not filled yet.
A linear scan of the disassembly, per kernel; works on hipcc -S output and on `llvm-objdump -d` of a code object.  tests/test_code_objects.py runs it.

    python tools/check_held_loads.py <file.s | file.dis> <kernel-name-substring> ...
"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def functions(text):
    """[(name, [instruction lines])] of an assembly listing or a disassembly."""
    out, name, body = [], None, []
    for l in text.split("\n"):
        m = re.match(r"^(?:[0-9a-f]+ <)?(_Z\w+)>?:", l)
        if m:
            if name:
                out.append((name, body))
            name, body = m.group(1), []
            continue
        u = l.split("//")[0].strip()
        if name is None or not u or u.startswith(";") or u.startswith("."):
            if u.startswith(".Lfunc_end") and name:
                out.append((name, body))
                name, body = None, []
            continue
        body.append(u)
    if name:
        out.append((name, body))
    return out


def check_text(text, wanted):
    bad, seen, loads = [], 0, 0
    for name, body in functions(text):
        if not any(w in name for w in wanted):
            continue
        t = re.search(r"ILi(\d+)ELi(\d+)E", name)
        V = int(t.group(1)) * int(t.group(2)) // 4 if t else 0
        seen += 1
        for j, u in enumerate(body):
            if not (u.startswith("global_load_dword") and "sc1" in u and "lds" not in u):
                continue
            loads += 1
            dst = regs(u.split()[1].rstrip(","))
            for w in body[j + 1:]:
                if w.startswith("s_waitcnt") and ("vmcnt(0)" in w or (V and f"vmcnt({V})" in w)):
                    break
                if w.startswith("s_endpgm"):
                    break
                if w.startswith("global_load") and "sc1" in w:
                    continue
                used = set()
                for tk in re.findall(r"v\[\d+:\d+\]|v\d+", w):
                    used |= regs(tk)
                if used & dst:
                    bad.append((name, u, w))
                    break
    return seen, loads, bad


def check(path, wanted):
    return check_text(open(path).read(), wanted)


if __name__ == "__main__":
    seen, loads, bad = check(sys.argv[1], sys.argv[2:])
    for b in bad:
        print("OFFENDER", *b, sep="\n    ")
    print(f"{seen} kernels, {loads} poll loads, {len(bad)} offending")
    sys.exit(1 if bad else 0)
