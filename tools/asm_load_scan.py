"""Do the hand-placed weight loads of the head kernel stay untouched until their wait?

The feature-split kernels request weight fragments with inline-asm `global_load_dwordx4` and wait for them with a
hand-written `s_waitcnt vmcnt(0)`: the compiler does not know the registers are in flight, so code IT places between
the load and the wait must not read, move or overwrite them.  This scan walks the kernels of a built library whose name
matches (default: head32_kernel) in program order, keeps the vector-memory events in flight in issue order (a
`s_waitcnt vmcnt(N)` retires all but the N youngest) and reports every instruction that names a register a
`global_load_dwordx4 v|a[..], v, s[..]` still in flight is going to write.

Control flow: the walk is linear.  Behind an unconditional forward `s_branch` the walk continues with nothing in flight
(the code there is reached from elsewhere, in a state the walk does not know) and picks the state at the branch up
again at its target.

    python tools/asm_load_scan.py [library.so] [kernel-name-substring]      exit 1 if anything is reported
"""
import os
import re
import subprocess
import sys
import tempfile
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
LOAD = re.compile(r'^\s*global_load_dwordx4\s+([va])\[(\d+):(\d+)\],\s*(?:v\d+,\s*s\[\d+:\d+\]|v\[\d+:\d+\],\s*off)')
REG = re.compile(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def listings(path):
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, local)
        subprocess.run([f'{LLVM}/llvm-objdump', '--offloading', local], check=True, cwd=tmp, capture_output=True)
        for obj in sorted(os.listdir(tmp)):
            if 'gfx950' in obj:
                yield subprocess.run([f'{LLVM}/llvm-objdump', '-d', os.path.join(tmp, obj)], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp)


def scan(path, want):
    """walk each matching kernel in program order; vector-memory events retire in issue order (gfx9: loads and stores
    share vmcnt), `vmcnt(N)` retires all but the N youngest"""
    bad, seen, tracked = [], 0, 0
    for listing in listings(path):
        name, inflight, done = None, [], True
        resume = {}                                             # branch target address -> the state at the branch
        for line in listing.splitlines():
            m = re.match(r'^[0-9a-f]+ <(.*)>:$', line)
            if m:
                name, inflight, done, resume = m.group(1), [], want not in m.group(1), {}
                seen += not done
                continue
            if done or name is None:
                continue
            text = line.split('//')[0]
            am = re.search(r'//\s*([0-9A-Fa-f]+):', line)
            addr = int(am.group(1), 16) if am else None
            if addr is not None and addr in resume:
                inflight = resume.pop(addr) + [e for e in inflight if e]
            b = re.match(r'^\s*s_branch\s+(\d+)', text)
            if b and addr is not None and int(b.group(1)) < 32768:
                resume[addr + 4 + 4 * int(b.group(1))] = inflight
                inflight = []
                continue
            w = re.search(r's_waitcnt\s+.*vmcnt\((\d+)\)', text)
            if w:
                keep = int(w.group(1))
                inflight = inflight[len(inflight) - keep:] if keep else []
                continue
            m = LOAD.match(text)
            if m:
                inflight.append({(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)})
                tracked += 1
                continue
            if re.match(r'^\s*(global|buffer|flat|scratch)_(load|store|atomic)', text):
                inflight.append(set())                      # any other vector-memory event: counted (gfx9: one counter, in order)
                continue
            live = set().union(*inflight) if inflight else set()
            hit = regs_of(text) & live
            if hit:
                bad.append((name, text.strip(), sorted(hit)[:4]))
    return seen, tracked, bad


if __name__ == '__main__':
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'ppgs_amd', 'libppgs_amd.so')
    want = sys.argv[2] if len(sys.argv) > 2 else 'head32_kernel'
    seen, tracked, bad = scan(lib, want)
    print(f'{seen} kernels matching "{want}" in {os.path.basename(lib)}, {tracked} hand-placed loads followed: {len(bad)} instructions touch a register in flight')
    for name, text, hit in bad[:40]:
        print(f'  {name[:60]}: {text}   {hit}')
    sys.exit(1 if bad or not seen else 0)
