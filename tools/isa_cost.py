"""Issue-cost budget of a loop of the device ISA, per basic block: instructions by class x the issue cost
measured by tools/valu_rates.hip (profiles/r05_valu_issue_rates.txt, MI355X, 4 wavefronts per SIMD:
cycles per instruction per SIMD).

    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -S --cuda-device-only -o /tmp/k.s <file with the kernel>
    python tools/isa_cost.py /tmp/k.s <kernel name fragment> <first label> <last label> [label=weight ...]

Prints every basic block between the two labels with its instruction classes; `label=weight` gives the
number of times a block runs per loop iteration (default 1; 0 drops a block: error paths, refill loops,
the other cases of a switch), and the weighted total."""
import re
import sys

COST = {'f64': 4.5, 'cmp': 4.5, 'vop3_32': 4.7, 'dpp': 5.0, 'vop12_32': 2.8, 'salu': 0.0, 'lds': 0.0, 'vmem': 0.0,
        'branch': 0.0, 'nop': 0.0, 'other': 0.0}
VOP3_ONLY = ('v_bitop3', 'v_lshl_add', 'v_lshl_or', 'v_readlane', 'v_writelane', 'v_mad_', 'v_bfe', 'v_bfi', 'v_perm',
             'v_alignbit', 'v_and_or', 'v_or3', 'v_add3', 'v_lshrrev_b64', 'v_lshlrev_b64', 'v_ashrrev_i64',
             'v_mbcnt', 'v_cvt_pk', 'v_mul_lo', 'v_mul_hi', 'v_min3', 'v_max3', 'v_med3', 'v_fma_f32')


def classify(ins):
    op = ins.split()[0]
    if op.startswith('s_nop') or op.startswith('s_waitcnt'):
        return 'nop'
    if op.startswith('s_cbranch') or op.startswith('s_branch') or op.startswith('s_setpc'):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if not op.startswith('v_'):
        return 'other'
    if op.startswith('v_cmp') or op.startswith('v_cmpx'):
        return 'cmp'
    if '_dpp' in op or ' quad_perm' in ins or ' row_' in ins or ' wave_sh' in ins:
        return 'dpp'
    if op.endswith('_f64') or '_f64_' in op or op.startswith('v_mov_b64') or 'f64' in op:
        return 'f64'
    if op.endswith('_e64') or op.startswith(VOP3_ONLY):
        return 'vop3_32'
    return 'vop12_32'


def main():
    text = open(sys.argv[1]).read()
    frag, first, last = sys.argv[2], sys.argv[3], sys.argv[4]
    weights = {}
    for a in sys.argv[5:]:
        k, v = a.split('=')
        weights[k] = float(v)
    m = re.search(r'\n(_Z\w*%s\w*):' % re.escape(frag), text)
    body = text[m.start():text.find('.Lfunc_end', m.start())].split('\n')
    blocks, cur, name = [], [], 'entry'
    for l in body:
        mm = re.match(r'(\.LBB\d+_\d+):', l)
        if mm:
            blocks.append((name, cur))
            name, cur = mm.group(1), []
            continue
        mm = re.match(r'; %bb\.(\d+):', l.strip())
        if mm:
            blocks.append((name, cur))
            name, cur = 'bb.' + mm.group(1), []
            continue
        t = l.strip()
        if t and not t.startswith((';', '.')) and not t.endswith(':'):
            cur.append(t)
    blocks.append((name, cur))
    names = [b[0] for b in blocks]
    i0, i1 = names.index(first), names.index(last)
    tot = dict.fromkeys(COST, 0.0)
    print('%-12s %6s  ' % ('block', 'weight') + ' '.join('%8s' % k for k in COST) + '   cycles')
    for name, ins in blocks[i0:i1 + 1]:
        w = weights.get(name, 1.0)
        c = dict.fromkeys(COST, 0)
        for x in ins:
            if x.startswith(';;#ASM') or x.startswith(';'):
                continue
            c[classify(x)] += 1
        cyc = sum(c[k] * COST[k] for k in COST)
        print('%-12s %6.2f  ' % (name, w) + ' '.join('%8d' % c[k] for k in COST) + '   %7.1f' % (cyc * w))
        for k in COST:
            tot[k] += w * c[k]
    print('%-12s %6s  ' % ('weighted sum', '') + ' '.join('%8.1f' % tot[k] for k in COST) +
          '   %7.1f' % sum(tot[k] * COST[k] for k in COST))
    print('VALU instructions (weighted): %.1f' % sum(tot[k] for k in ('f64', 'cmp', 'vop3_32', 'dpp', 'vop12_32')))


if __name__ == '__main__':
    main()
