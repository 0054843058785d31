"""Condensed view of one kernel's assembly: per basic block, runs of buffer loads (L), stores (S),
scratch accesses (sl/ss), vmcnt waits (Wn) and the number of vector ALU instructions between them.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only ... -o mpcx.s mpcx_kernels.hip
    python scripts/isa_trace.py mpcx.s _ZN4mpcx13k_solve_stageILi1EEEvNS_9SolveArgsEPd [min_loads]
Shows where a loop body pays more than one memory round trip per trip."""
import re, sys

path, sym = sys.argv[1], sys.argv[2]
min_loads = int(sys.argv[3]) if len(sys.argv) > 3 else 6
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ["entry:"]
run, n, valu = None, 0, 0
def flush():
    global run, n, valu
    if valu: cur.append(f"v{valu}"); valu = 0
    if run: cur.append(f"{run}x{n}")
    run, n = None, 0
for ln in lines[start + 1:end]:
    s = ln.strip()
    if not s or s.startswith(';'): continue
    if s.startswith('.LBB'):
        flush(); blocks.append(cur); cur = [s.split(':')[0] + ":"]; continue
    if s.startswith('.'): continue
    op = s.split()[0]
    k = None
    if op.startswith('buffer_load'): k = 'L'
    elif op.startswith('buffer_store'): k = 'S'
    elif op.startswith('scratch_load'): k = 'sl'
    elif op.startswith('scratch_store'): k = 'ss'
    elif op.startswith('global_load'): k = 'gl'
    elif op.startswith('global_store'): k = 'gs'
    elif op == 's_waitcnt':
        m = re.search(r'vmcnt\((\d+)\)', s)
        if m: k = 'W' + m.group(1)
    elif op.startswith('s_cbranch') or op == 's_branch':
        flush(); cur.append(op[2:] + "->" + s.split()[-1]); continue
    if k is None:
        if op.startswith('v_'):
            if run: flush()
            valu += 1
        continue
    if k != run: flush(); run, n = k, 1
    else: n += 1
flush(); blocks.append(cur)
for b in blocks:
    t = ' '.join(b)
    nl = sum(int(x) for x in re.findall(r'\bLx(\d+)', t))
    if nl >= min_loads: print(f"[loads {nl}]", t, "\n")
