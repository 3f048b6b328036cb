"""Per-loop summary of the device ISA: for every innermost loop of the named kernels, the
instructions, global loads / stores, LDS operations and barriers of one iteration.  A streaming
loop with ONE global load per iteration keeps one request per thread in flight (latency bound).

  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -S --cuda-device-only \
        -o /tmp/eng.s tombo_amd/csrc/tba_engine.hip
  python tools/isa_loops.py /tmp/eng.s k_normalize k_rescale_absz ...
"""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    want = sys.argv[2:]
    for m in re.finditer(r'\n(_Z\w+):\s*; @', s):
        n = m.group(1)
        if want and not any(w in n for w in want):
            continue
        i = m.start()
        j = s.find('.Lfunc_end', i)
        body = s[i:j].split('\n')
        blocks, cur = [], []
        for l in body:
            if re.match(r'\.LBB\d+_\d+:', l):
                blocks.append(cur)
                cur = [l]
            else:
                cur.append(l)
        blocks.append(cur)
        k = s.find('.set %s.num_vgpr, ' % n)
        vg = s[k:k + 200].split(', ')[1].split('\n')[0] if k >= 0 else '?'
        print('%s  vgpr %s' % (n[:70], vg))
        for b in blocks:
            if b and 'Inner Loop Header' in b[0]:
                txt = '\n'.join(b)
                gl = txt.count('global_load')
                if gl:
                    print('    %-12s insts %4d  gload %2d  gstore %2d  lds %3d  barrier %d' % (
                        b[0].split(':')[0], len(b), gl, txt.count('global_store'),
                        txt.count('\tds_'), txt.count('s_barrier')))


if __name__ == '__main__':
    main()
