"""Build a library whose DEVICE code is hand-edited ISA: the device assembly of a build (hipcc -S --cuda-device-only)
is patched inside k_main_tb_par<16>, assembled, bundled and embedded under the unchanged host code
(round-6 fault hunt: which instruction pair of the failing build misbehaves; profiles/r06_traceback_rootcause.txt).

    python tools/asm_variant.py <base.s> <out.so> <variant> [extra -D flags of the base build ...]
"""
import os
import re
import subprocess
import sys

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
KERNEL = '_Z13k_main_tb_parILi16E'


def edit(lines, variant):
    a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and ':' in l)
    b = next(i for i in range(a, len(lines)) if lines[i].startswith('.Lfunc_end'))
    inv = [i for i in range(a, b) if 'buffer_inv sc1' in lines[i]]
    assert len(inv) in (1, 2), inv
    pa, pb = inv[0], (inv[1] if len(inv) == 2 else b)   # phase B = behind the fence that follows phase A
    out, n = [], 0
    m = re.fullmatch(r'desc_v(\d+)_a(\d+)_s(\d+)', variant)
    if m:   # kernel descriptor only: next_free_vgpr, accum_offset, next_free_sgpr
        d = next(i for i, l in enumerate(lines) if l.strip().startswith('.amdhsa_kernel ' + KERNEL))
        for i in range(d, d + 40):
            lines[i] = lines[i].replace('.amdhsa_next_free_vgpr 224', '.amdhsa_next_free_vgpr ' + m.group(1)).replace(
                '.amdhsa_accum_offset 224', '.amdhsa_accum_offset ' + m.group(2)).replace('.amdhsa_next_free_sgpr 40', '.amdhsa_next_free_sgpr ' + m.group(3))
        return lines, 0
    if variant.startswith('L1_'):
        k = next(i for i in range(pa, pb) if re.match(r'\s*v_subbrev_co_u32_e32 v64, vcc, 0, v215, vcc', lines[i]))
        smem = ['\ts_getpc_b64 s[40:41]', '\ts_add_u32 s40, s40, TBA_TB_B2_OFF@rel32@lo+4', '\ts_addc_u32 s41, s41, TBA_TB_B2_OFF@rel32@hi+12',
                '\ts_load_dwordx2 s[40:41], s[40:41], 0x0']
        movs = ['\tv_lshl_add_u64 v[224:225], v[88:89], 3, v[72:73]', '\tv_mov_b32_e32 v226, v216', '\tv_mov_b32_e32 v227, v215',
                '\tv_mov_b32_e32 v228, v64', '\tv_mov_b32_e32 v229, vcc_lo', '\tv_mov_b32_e32 v230, s2', '\tv_mov_b32_e32 v231, s3']
        ins = {'L1_nop16': ['\ts_nop 7', '\ts_nop 7'], 'L1_lgkm0': ['\ts_waitcnt lgkmcnt(0)'], 'L1_vm0': ['\ts_waitcnt vmcnt(0)'],
               'L1_none_desc': [], 'L1_nop0_nodesc': ['\ts_nop 0'], 'L1_lgkm0_nodesc': ['\ts_waitcnt lgkmcnt(0)'], 'L1_nop16_nodesc': ['\ts_nop 7', '\ts_nop 7'],
               'L1_movs': movs, 'L1_smem': smem + ['\ts_waitcnt lgkmcnt(0)'],
               'L1_nostores': smem + movs + ['\ts_waitcnt lgkmcnt(0)', '\ts_lshl_b64 s[40:41], s[40:41], 4', '\tv_lshl_add_u64 v[224:225], s[40:41], 0, v[224:225]']}[variant]
        lines = lines[:k + 1] + ins + lines[k + 1:]
        d = next(i for i, l in enumerate(lines) if l.strip().startswith('.amdhsa_kernel ' + KERNEL))
        for i in range(d, d + 40 if not variant.endswith('_nodesc') else d):
            lines[i] = lines[i].replace('.amdhsa_next_free_vgpr 224', '.amdhsa_next_free_vgpr 232').replace(
                '.amdhsa_accum_offset 224', '.amdhsa_accum_offset 232').replace('.amdhsa_next_free_sgpr 40', '.amdhsa_next_free_sgpr 42')
        return lines, len(ins)
    if variant == 'trace_m':
        # row 0 of phase B: what the move selection produced, stored into the B2 build's third array under the
        # entry state (slots lo-1 .. lo-3 of the boundary): m | bp << 32, bp' | vcc_lo << 32, s[2:3]
        k = next(i for i in range(pa, pb) if re.match(r'\s*v_subbrev_co_u32_e32 v64, vcc, 0, v215, vcc', lines[i]))
        ins = """	s_getpc_b64 s[40:41]
	s_add_u32 s40, s40, TBA_TB_B2_OFF@rel32@lo+4
	s_addc_u32 s41, s41, TBA_TB_B2_OFF@rel32@hi+12
	s_load_dwordx2 s[40:41], s[40:41], 0x0
	v_lshl_add_u64 v[224:225], v[88:89], 3, v[72:73]
	v_mov_b32_e32 v226, v216
	v_mov_b32_e32 v227, v215
	v_mov_b32_e32 v228, v64
	v_mov_b32_e32 v229, vcc_lo
	v_mov_b32_e32 v230, s2
	v_mov_b32_e32 v231, s3
	s_waitcnt lgkmcnt(0)
	s_lshl_b64 s[40:41], s[40:41], 4
	v_lshl_add_u64 v[224:225], s[40:41], 0, v[224:225]
	global_store_dwordx2 v[224:225], v[226:227], off offset:-8
	global_store_dwordx2 v[224:225], v[228:229], off offset:-16
	global_store_dwordx2 v[224:225], v[230:231], off offset:-24""".split('\n')
        lines = lines[:k + 1] + ins + lines[k + 1:]
        lines = [l.replace('.amdhsa_next_free_vgpr 224', '.amdhsa_next_free_vgpr 232').replace('.amdhsa_accum_offset 224', '.amdhsa_accum_offset 232')
                 .replace('.amdhsa_next_free_sgpr 40', '.amdhsa_next_free_sgpr 42') if a < 0 else l for l in lines]
        # (the kernel descriptor of this kernel only)
        d = next(i for i, l in enumerate(lines) if l.strip().startswith('.amdhsa_kernel ' + KERNEL))
        for i in range(d, d + 40):
            lines[i] = lines[i].replace('.amdhsa_next_free_vgpr 224', '.amdhsa_next_free_vgpr 232').replace(
                '.amdhsa_accum_offset 224', '.amdhsa_accum_offset 232').replace('.amdhsa_next_free_sgpr 40', '.amdhsa_next_free_sgpr 42')
        return lines, len(ins)
    for i, l in enumerate(lines):
        in_b = pa < i < pb
        in_a = a < i < pa
        t = l.strip()
        if variant == 'after_cndmask_m' and in_b and re.match(r'v_cndmask_b32_e64 v\d+, 2, 1, s\[', t):
            out += [l, '\ts_nop 0']; n += 1; continue
        if variant == 'before_cndmask_m' and in_b and re.match(r'v_cndmask_b32_e64 v\d+, 2, 1, s\[', t):
            out += ['\ts_nop 0', l]; n += 1; continue
        if variant == 'after_cmp_m2' and in_b and re.match(r'v_cmp_eq_u32_e32 vcc, 2, v\d+', t):
            out += [l, '\ts_nop 1']; n += 1; continue
        if variant == 'shift4' and i == pa + 2:
            out += ['\ts_nop 0', l]; n += 1; continue
        if variant == 'shift4_a' and i == a + 3:
            out += ['\ts_nop 0', l]; n += 1; continue
        if variant == 'after_cndmask_m_a' and in_a and re.match(r'v_cndmask_b32_e64 v\d+, 2, 1, s\[', t):
            out += [l, '\ts_nop 0']; n += 1; continue
        if variant.startswith('after_every_cndmask_sgpr') and in_b and re.match(r'v_cndmask_b32_e64 .*, s\[\d+:\d+\]$', t):
            out += [l, '\ts_nop 0']; n += 1; continue
        out.append(l)
    return out, n


def main():
    base, so, variant = sys.argv[1:4]
    flags = sys.argv[4:]
    lines = open(base).read().split('\n')
    lines, n = edit(lines, variant) if variant != 'control' else (lines, 0)
    w = so[:-3] + '_work'
    os.makedirs(w, exist_ok=True)
    open(w + '/dev.s', 'w').write('\n'.join(lines))
    run = lambda c: subprocess.check_call(c)
    run([LLVM + '/clang', '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', w + '/dev.s', '-o', w + '/dev.o'])
    run([LLVM + '/ld.lld', '-shared', w + '/dev.o', '-o', w + '/dev.hsaco'])
    run([LLVM + '/clang-offload-bundler', '-type=o', '-bundle-align=4096',
         '-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950', '-input=/dev/null',
         '-input=' + w + '/dev.hsaco', '-output=' + w + '/dev.hipfb'])
    run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared'] + flags +
        ['--cuda-host-only', '-Xclang', '-fcuda-include-gpubinary', '-Xclang', w + '/dev.hipfb', '-o', so,
         os.path.join(ROOT, 'tombo_amd', 'csrc', 'tba_engine.hip')])
    subprocess.call(['rm', '-rf', w])
    print('%s: %d edits -> %s' % (variant, n, so))


if __name__ == '__main__':
    main()
