"""Reads of never-written vector registers in gfx950 device assembly (hipcc --cuda-device-only -S).

Why: round 4 met a hipcc (ROCm 7.2) build of k_g2_accum28 that STORED two limbs of the accumulator from registers no instruction of the kernel
writes on the path that skips the accumulation loop (g2.hip, the `asm volatile` pin).  This is the check that would have caught it without a GPU:
a forward reaching-definitions pass over the kernel's control-flow graph (labels / s_branch / s_cbranch_*), MAY-defined sets joined by union, and a
report of every instruction that reads a VGPR / AGPR with NO definition on ANY path from the kernel's entry.  Union at joins means no false positives
from divergent control flow (a register written under one exec mask and read under the same one later is fine); what it reports is certainly wrong
code (or an intended read of garbage, which this code base has none of).

  python3 tools/isa_undef_check.py [--all-reads] file.s [kernel-name-substring ...]      exit status 1 when something is reported
(default: memory writes only; --all-reads lists every instruction that reads such a register)
As a module: undefined_reads(asm_text, name_substrings) -> {kernel: [(line_no, instruction, [registers])]}."""
import re, sys

_REG = re.compile(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]')
_STORE = re.compile(r'^(global|flat|scratch|buffer)_store|^ds_(write|store)|^(global|flat|buffer|ds)_atomic|^ds_(add|sub|min|max|and|or|xor|inc|dec)_')
_NO_VDST = re.compile(r'^v_cmp|^v_readlane|^v_readfirstlane|^s_|^v_nop|^buffer_wbl2|^buffer_inv|^buffer_gl')
_RMW_DST = re.compile(r'^v_(fmac|mac|pk_fmac|dot\dc|fmaak|madak)|^v_mfma|^v_smfmac')      # the destination is also a source
_ENTRY = {('v', 0), ('v', 1), ('v', 2)}                    # work-item ids (packed into v0 on gfx950; v1 / v2 tolerated)


def _regs(text):
    out = []
    for m in _REG.finditer(text):
        if m.group(1): out.append((m.group(1), int(m.group(2))))
        else: out += [(m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1)]
    return out


def _split_operands(rest):
    rest = rest.split(';')[0].strip()
    return [o.strip() for o in rest.split(',')] if rest else []


def _defs_uses(op, operands):
    """(defined registers, read registers) of one instruction."""
    if _STORE.match(op):
        d = []
        if 'atomic' in op or op.startswith('ds_') and not (op.startswith('ds_write') or op.startswith('ds_store')):      # returning atomics name a destination first; treat operand 0 as both
            pass
        return d, [r for o in operands for r in _regs(o)]
    if _NO_VDST.match(op): return [], [r for o in operands for r in _regs(o)]
    if op.startswith('v_swap'):
        both = [r for o in operands[:2] for r in _regs(o)]
        return both, both
    if not operands: return [], []
    d = _regs(operands[0]); u = [r for o in operands[1:] for r in _regs(o)]
    if _RMW_DST.match(op): u += d
    return d, u


def _kernels(text):
    names = re.findall(r'^\s*\.amdhsa_kernel\s+(\S+)', text, re.M)
    return names


def _body(lines, name):
    start = None
    for i, l in enumerate(lines):
        if l.startswith(name + ':'): start = i + 1; break
    if start is None: return None, 0
    end = start
    while end < len(lines) and not lines[end].startswith('.Lfunc_end'): end += 1
    return lines[start:end], start


def undefined_reads(text, wanted=(), stores_only=True):
    """stores_only: report only memory writes (stores / atomics) that read a never-written register — the failure this check exists for.  The wider
    report (every instruction) also lists a benign compiler idiom: the low half of a 64-bit multiply-add computed with an undefined high-half addend
    (`v_mad_u64_u32 v[4:5], .., v[4:5]` with only v4 live), e.g. in k_gather_strided."""
    lines = text.split('\n'); report = {}
    for name in _kernels(text):
        if wanted and not any(w in name for w in wanted): continue
        body, base = _body(lines, name)
        if body is None: continue
        # basic blocks
        blocks = [[]]; label_of = {};
        for off, l in enumerate(body):
            t = l.strip()
            if not t or t.startswith(';') or t.startswith('.') and not re.match(r'^\.L\w+:', t): continue
            m = re.match(r'^(\.L\w+):', t)
            if m:
                if blocks[-1]: blocks.append([])
                label_of[m.group(1)] = len(blocks) - 1
                continue
            parts = t.split(None, 1); op = parts[0]; operands = _split_operands(parts[1] if len(parts) > 1 else '')
            blocks[-1].append((base + off + 1, op, operands, t))
            if op.startswith('s_branch') or op.startswith('s_cbranch') or op in ('s_endpgm', 's_setpc_b64'): blocks.append([])
        # labels that pointed at an empty trailing block index stay valid because blocks are only appended
        succ = []
        for bi, b in enumerate(blocks):
            s = []
            if b:
                _, op, operands, _t = b[-1]
                if op.startswith('s_branch'): s = [label_of[operands[0]]] if operands and operands[0] in label_of else []
                elif op.startswith('s_cbranch'):
                    if operands and operands[-1] in label_of: s.append(label_of[operands[-1]])
                    if bi + 1 < len(blocks): s.append(bi + 1)
                elif op in ('s_endpgm', 's_setpc_b64'): s = []
                elif bi + 1 < len(blocks): s = [bi + 1]
            elif bi + 1 < len(blocks): s = [bi + 1]
            succ.append(s)
        ALL = {(k, i) for k in 'va' for i in range(512)}
        inset = [None] * len(blocks); inset[0] = set(_ENTRY); work = [0]
        gen = []
        for b in blocks:
            g = set()
            for _, op, operands, _t in b:
                if op == 's_swappc_b64': g |= ALL                # a call: the callee may define anything
                d, _u = _defs_uses(op, operands); g |= set(d)
            gen.append(g)
        while work:
            bi = work.pop(); out = inset[bi] | gen[bi]
            for s in succ[bi]:
                if inset[s] is None: inset[s] = set(out); work.append(s)
                elif not out <= inset[s]: inset[s] |= out; work.append(s)
        found = []
        for bi, b in enumerate(blocks):
            if inset[bi] is None: continue                       # unreachable
            cur = set(inset[bi])
            for ln, op, operands, t in b:
                if op == 's_swappc_b64': cur |= ALL
                d, u = _defs_uses(op, operands)
                bad = sorted({r for r in u if r not in cur})
                if bad and (not stores_only or _STORE.match(op)): found.append((ln, t, ['%s%d' % r for r in bad]))
                cur |= set(d)
        report[name] = found
    return report


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if a != '--all-reads']
    txt = open(args[0]).read(); rep = undefined_reads(txt, args[1:], stores_only='--all-reads' not in sys.argv); n = 0
    for k, f in rep.items():
        print('%s: %d read(s) of never-written registers' % (k, len(f)))
        for ln, t, regs in f[:40]: print('   line %d: %-70s <- %s' % (ln, t[:70], ' '.join(regs)))
        n += len(f)
    sys.exit(1 if n else 0)
