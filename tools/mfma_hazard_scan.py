"""MFMA results in ARCHITECTURAL registers: is every consumer far enough behind the MFMA that wrote them?

The feature-split kernels issue some MFMAs from inline asm with the accumulator in architectural VGPRs
(ppg_device.h: mma32v0 / mma32v -- phase A of the layer kernels' chunk loop, whose h is packed by VALU instructions).
hipcc's hazard recognizer does not look inside asm: it inserts neither the wait states between an MFMA's write and a
VALU / LDS / memory instruction that reads or overwrites the result, nor the ones between an MFMA still reading its C
operand and a VALU write of those registers.  The distance exists only as source placement, and the scheduler may move
plain C++ consumers across `asm volatile`.  This scan audits the BUILT code:

  * RAW / WAW: from a `v_mfma_*` whose destination is v[..] to the first later instruction that names a destination
    register -- unless that instruction is an MFMA accumulating into the same registers (the matrix pipe forwards its own
    chain) -- there must be at least PASSES + 4 wait states (gfx940 ISA guide, "XDL write VGPR -> VALU read / write,
    VMEM / LDS read, XDL read as SrcA / B": 2 passes 5, 4: 7, 8: 11, 16: 19; one more kept in hand for gfx950);
  * WAR: a VALU / load write of registers an MFMA reads as its C operand (C != destination) must be PASSES - 1 wait
    states behind it (gfx940: 8 passes 7, 16 passes 15).

Scope: the feature-split kernels (the kernels that may use all 512 registers: there every BUILTIN MFMA gets the
accumulation-register form, so an MFMA with its result in v[..] is one of the asm ones); elsewhere the compiler places
and guards its own MFMAs and this walk would only second-guess its hazard recognizer.

Wait states: an `s_nop N` counts N + 1, an intervening MFMA its own passes (the pipe takes the next one only then), any
other instruction 1.  The walk is linear (branches are not followed; the windows are a few instructions long).

    python tools/mfma_hazard_scan.py [library.so] [kernel-name-substring ...]   exit 1 if anything is reported
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_load_scan import listings, regs_of          # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MFMA = re.compile(r'^\s*(v_mfma_\w+)\s+(.*)$')
# passes of the MFMA shapes this library issues (4 cycles each)
PASSES = {'32x32x16': 8, '16x16x32': 4, '32x32x8': 16, '16x16x16': 8, '16x16x4': 8, '32x32x2': 16, '4x4x4': 2, '16x16x64': 4, '32x32x32': 8}
MARGIN = 4


def passes_of(op):
    m = re.search(r'_(\d+x\d+x\d+)', op)
    return PASSES.get(m.group(1), 16) if m else 16


def operands(text):
    return [t.strip() for t in re.split(r',\s*(?![^\[]*\])', text.split(' cbsz')[0].split(' abid')[0].split(' blgp')[0])]


FAMILIES = ('layer32_kernel', 'ffn32x2_kernel', 'head32_kernel', 'gemm32_kernel', 'posconv_kernel')


def scan(path, want=FAMILIES):
    """-> (kernels seen, MFMAs with an architectural destination, findings)"""
    bad, seen, audited = [], 0, 0
    for listing in listings(path):
        name, live = None, []
        for line in listing.splitlines():
            m = re.match(r'^[0-9a-f]+ <(.*)>:$', line)
            if m:
                name, live = (m.group(1) if any(w in m.group(1) for w in want) else None), []
                seen += name is not None
                continue
            if name is None:
                continue
            text = line.split('//')[0].strip()
            if not text or text.endswith(':'):
                continue
            mm = MFMA.match(text)
            nop = re.match(r'^s_nop\s+(\d+)', text)
            named = regs_of(text)
            # 1. this instruction against the MFMAs still inside their windows
            still = []
            for e in live:
                touched = named & e['dst']
                chained = mm is not None and e['dst'] == regs_of(operands(mm.group(2))[0]) and e['dst'] == regs_of(operands(mm.group(2))[-1])
                if touched and not chained:
                    if e['waited'] < e['need']:
                        bad.append((name, e['text'], text, f"{e['waited']} wait states of {e['need']} (result)"))
                    continue                                   # first consumer seen: this MFMA's result is accounted for
                if chained:
                    continue                                   # the chain's next MFMA takes over below
                if e['srcc'] and not mm and text.startswith('v_') and ' ' in text:
                    # a VALU write of the C operand's registers while the MFMA may still read them (loads return far later)
                    dest = regs_of(operands(text.split(None, 1)[1])[0])
                    if dest & e['srcc'] and e['waited'] < e['need_war']:
                        bad.append((name, e['text'], text, f"{e['waited']} wait states of {e['need_war']} (C operand overwritten)"))
                e['waited'] += (int(nop.group(1)) + 1) if nop else (passes_of(mm.group(1)) if mm else 1)
                if e['waited'] < max(e['need'], e['need_war']):
                    still.append(e)
            live = still
            # 2. a new MFMA with an architectural destination
            if mm:
                ops = operands(mm.group(2))
                if ops and ops[0].startswith('v'):
                    p = passes_of(mm.group(1))
                    dst, srcc = regs_of(ops[0]), regs_of(ops[-1]) if ops[-1].startswith('v') else set()
                    live.append({'dst': dst, 'srcc': srcc - dst, 'need': p + MARGIN, 'need_war': p - 1, 'waited': 0, 'text': text})
                    audited += 1
    return seen, audited, bad


if __name__ == '__main__':
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'ppgs_amd', 'libppgs_amd.so')
    want = tuple(sys.argv[2:]) or FAMILIES
    seen, audited, bad = scan(lib, want)
    print(f'{seen} kernels, {audited} MFMAs with the result in architectural registers, {len(bad)} findings')
    for b in bad[:40]:
        print('  ', b)
    sys.exit(1 if bad else 0)
