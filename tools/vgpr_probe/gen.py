# generates vgpr_probe_<alloc>.s : every wave fills v0..v(NV-1) with a signature, idles a wave-dependent time while
# neighbours come and go, then checks every register (all lanes hold the same value: v_readfirstlane + scalar compare)
import os
import sys
NV = int(sys.argv[1]); ALLOC = int(sys.argv[2]); name = sys.argv[3]
L = []
A = L.append
A('\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
A('\t.amdhsa_code_object_version 6')
A('\t.text')
A('\t.protected %s' % name); A('\t.globl %s' % name); A('\t.p2align 8'); A('\t.type %s,@function' % name)
A('%s:' % name)
# s[0:1] kernarg ptr, s2 = workgroup id x
A('\ts_load_dwordx2 s[4:5], s[0:1], 0x0')      # out
A('\ts_load_dword s6, s[0:1], 0x8')           # base sleep iterations
A('\ts_lshl_b32 s8, s2, 10')                   # signature base = wg << 10
A('\ts_or_b32 s8, s8, 0x40000000')
for n in range(NV):
    A('\tv_mov_b32_e32 v%d, s8' % n)
    A('\tv_add_u32_e32 v%d, %d, v%d' % (n, n, n))
A('\ts_waitcnt lgkmcnt(0)')
if os.environ.get('LOADS'):
    A('\tv_mbcnt_lo_u32_b32 v159, -1, 0'); A('\tv_mbcnt_hi_u32_b32 v159, -1, v159'); A('\tv_lshlrev_b32_e32 v159, 4, v159')
    A('\ts_and_b32 s16, s2, 1023'); A('\ts_lshl_b32 s16, s16, 10'); A('\tv_add_u32_e32 v159, s16, v159')
    A('\ts_add_u32 s14, s4, 0x1000000'); A('\ts_addc_u32 s15, s5, 0')
# idle: (wg & 7) * base + base iterations of s_sleep
A('\ts_and_b32 s9, s2, 7'); A('\ts_add_u32 s9, s9, 1'); A('\ts_mul_i32 s9, s9, s6')
A('\ts_mov_b32 s20, -1'); A('\ts_mov_b32 s21, 0'); A('\ts_mov_b32 s22, 0'); A('\ts_mov_b64 s[24:25], 0'); A('\ts_mov_b32 s13, 0')
A('.Lidle_%s:' % name)
import os
if os.environ.get('MASKS'):
    # the failing decision's shape (k_tb_par.h: m = m2 ? 2 : 1 through an SGPR-pair mask a VALU compare has just written,
    # the result in one of the allocation's top registers, read a few instructions later): compare -> mask -> select -> check,
    # alternately all-true and all-false, results in v216..v223
    for n in range(0, 200, 2):
        t = 216 + (n // 2) % 8
        A('\tv_cmp_eq_u32_e64 s[26:27], v%d, v%d' % (n, n))          # all ones
        A('\tv_mov_b32_e32 v215, v%d' % (n + 1))
        A('\ts_mov_b32 s28, -1')
        A('\tv_cndmask_b32_e64 v%d, 1, 2, s[26:27]' % t)              # -> 2
        A('\ts_and_b64 s[26:27], s[26:27], exec')
        A('\tv_cmp_eq_u32_e32 vcc, 2, v%d' % t)
        A('\ts_xor_b64 s[30:31], s[28:29], -1')
        A('\tv_mov_b32_e32 v214, 11')
        A('\tv_subbrev_co_u32_e32 v213, vcc, 0, v215, vcc')           # v215 - 1 when the select gave 2
        A('\tv_sub_u32_e32 v213, v215, v213')                         # must be 1
        A('\tv_cmp_ne_u32_e32 vcc, 1, v213')
        A('\ts_or_b64 s[24:25], s[24:25], vcc')
        A('\tv_cmp_ne_u32_e64 s[26:27], v%d, v%d' % (n, n))          # all zeros
        A('\tv_mov_b32_e32 v215, v%d' % (n + 1))
        A('\ts_mov_b32 s28, -1')
        A('\tv_cndmask_b32_e64 v%d, 1, 2, s[26:27]' % t)              # -> 1
        A('\tv_cmp_ne_u32_e32 vcc, 1, v%d' % t)
        A('\ts_or_b64 s[24:25], s[24:25], vcc')
elif os.environ.get('LOADS'):
    # sixteen 16-byte loads into v[160:223] per iteration from a buffer of 0x5a5a5a5a words (the out buffer's tail, filled by
    # the host), the registers zeroed before: a load whose data does not arrive, or arrives elsewhere, shows in the compare
    A('\ts_add_u32 s13, s13, 1')
    for n in range(160, 224):
        A('\tv_mov_b32_e32 v%d, 0' % n)
    for k in range(16):
        A('\tglobal_load_dwordx4 v[%d:%d], v159, s[14:15] offset:%d' % (160 + 4 * k, 163 + 4 * k, 0))
    A('\ts_waitcnt vmcnt(0)')
    for n in range(160, 224):
        A('\tv_cmp_ne_u32_e32 vcc, 0x5a5a5a5a, v%d' % n)
        A('\ts_or_b64 s[24:25], s[24:25], vcc')
elif os.environ.get('WRITES'):
    # every register is incremented in every iteration (a write that is lost shows at the end: the active form only
    # rewrites a register with itself); s13 counts the iterations this wave really did
    A('\ts_add_u32 s13, s13, 1')
    for n in range(NV):
        A('\tv_add_u32_e32 v%d, 1, v%d' % (n, n))
elif os.environ.get('ACTIVE'):
    for n in range(NV):
        A('\ts_add_u32 s11, s8, %d' % n)
        A('\tv_cmp_ne_u32_e32 vcc, s11, v%d' % n)
        A('\tv_mov_b32_e32 v%d, v%d' % (n, n))
        A('\ts_or_b64 s[24:25], s[24:25], vcc')
else:
    A('\ts_sleep 20')
A('\ts_sub_u32 s9, s9, 1'); A('\ts_cmp_lg_u32 s9, 0'); A('\ts_cbranch_scc1 .Lidle_%s' % name)
A('\ts_cmp_eq_u64 s[24:25], 0'); A('\ts_cbranch_scc1 .Lclean_%s' % name); A('\ts_mov_b32 s22, 0x10000'); A('.Lclean_%s:' % name)
for n in range(NV if not (os.environ.get('LOADS') or os.environ.get('MASKS')) else (159 if os.environ.get('LOADS') else 200)):
    A('\tv_readfirstlane_b32 s10, v%d' % n)
    A('\ts_add_u32 s11, s8, %d' % n)
    if os.environ.get('WRITES'):
        A('\ts_add_u32 s11, s11, s13')
    A('\ts_cmp_eq_u32 s10, s11')
    A('\ts_cbranch_scc1 .Lok_%s_%d' % (name, n))
    A('\ts_add_u32 s22, s22, 1')
    A('\ts_cmp_lg_u32 s20, -1')
    A('\ts_cbranch_scc1 .Lok_%s_%d' % (name, n))
    A('\ts_mov_b32 s20, %d' % n); A('\ts_mov_b32 s21, s10')
    A('.Lok_%s_%d:' % (name, n))
# write s20,s21,s22,HW_ID to out[wg*4..] by lane 0
A('\ts_getreg_b32 s23, hwreg(HW_REG_HW_ID)')
A('\tv_mov_b32_e32 v0, s20'); A('\tv_mov_b32_e32 v1, s21'); A('\tv_mov_b32_e32 v2, s22'); A('\tv_mov_b32_e32 v3, s23')
A('\ts_lshl_b32 s12, s2, 4'); A('\tv_mov_b32_e32 v4, s12')
A('\ts_mov_b64 exec, 1')
A('\tglobal_store_dwordx4 v4, v[0:3], s[4:5]')
A('\ts_endpgm')
A('.Lfunc_end_%s:' % name)
A('\t.size %s, .Lfunc_end_%s-%s' % (name, name, name))
A('\t.rodata'); A('\t.p2align 6')
A('\t.amdhsa_kernel %s' % name)
for k, v in [('group_segment_fixed_size', 0), ('private_segment_fixed_size', 0), ('kernarg_size', 16),
             ('user_sgpr_count', 2), ('user_sgpr_kernarg_segment_ptr', 1), ('system_sgpr_workgroup_id_x', 1),
             ('system_vgpr_workitem_id', 0), ('next_free_vgpr', ALLOC), ('next_free_sgpr', 32), ('accum_offset', ALLOC if ALLOC % 4 == 0 else (ALLOC + 3) // 4 * 4),
             ('reserve_vcc', 1), ('float_round_mode_32', 0), ('float_round_mode_16_64', 0), ('float_denorm_mode_32', 3),
             ('float_denorm_mode_16_64', 3), ('dx10_clamp', 1), ('ieee_mode', 1), ('tg_split', 0)]:
    A('\t\t.amdhsa_%s %s' % (k, v))
A('\t.end_amdhsa_kernel')
A('\t.text')
A('\t.amdgpu_metadata')
A('''---
amdhsa.kernels:
  - .args:
      - .address_space: global
        .offset: 0
        .size: 8
        .value_kind: global_buffer
      - .offset: 8
        .size: 4
        .value_kind: by_value
    .group_segment_fixed_size: 0
    .kernarg_segment_align: 8
    .kernarg_segment_size: 16
    .max_flat_workgroup_size: 64
    .name: %s
    .private_segment_fixed_size: 0
    .sgpr_count: 40
    .symbol: %s.kd
    .vgpr_count: %d
    .wavefront_size: 64
amdhsa.target: amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...''' % (name, name, ALLOC))
A('\t.end_amdgpu_metadata')
open('%s.s' % name, 'w').write('\n'.join(L) + '\n')
