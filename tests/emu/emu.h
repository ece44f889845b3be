// gfx950emu -- TEST INFRASTRUCTURE ONLY (like oracle/): a functional, ISA-level emulator of the gfx950 code objects inside
// libflacgpu.so, behind a stand-in for the HIP runtime (tests/emu/libamdhip64.so.7).  The tests under tests/test_emu_*.py load it
// IN FRONT OF the product library in a child process, so that the library's own host code launches its own compiled kernels --
// the instructions hipcc emitted, inline assembly included -- on an interpreter instead of a GPU, and compare the results with
// the CPU oracle.  Nothing under pyflac_amd/ knows about it; the product has no CPU path (README).
//
// What it models: wave64 execution of the instructions the kernels use (SALU, VALU incl. SDWA / DPP / fp64 / v_mfma_f64_4x4x4,
// SMEM, global / flat / scratch memory, LDS incl. LDS-DMA, atomics), EXEC / VCC / SCC / M0, s_barrier, workgroups dispatched in
// order up to a residency cap, kernels of several HIP streams interleaved wave by wave (so kernels that wait for words other
// kernels raise make progress), HIP events, a 100 MHz wall clock derived from the instructions executed.
// What it does not model: caches and the memory model (memory is one coherent array), timing, hazards / wait states.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include <unordered_map>

typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64;
typedef int8_t i8; typedef int16_t i16; typedef int32_t i32; typedef int64_t i64;

enum OpndKind : u8 { K_NONE, K_SGPR, K_VGPR, K_AGPR, K_VCC, K_VCC_LO, K_VCC_HI, K_EXEC, K_EXEC_LO, K_EXEC_HI, K_M0, K_SCC, K_IMM, K_FIMM,
                     K_OFF, K_SHARED_BASE, K_PRIVATE_BASE, K_SHARED_LIMIT, K_PRIVATE_LIMIT, K_NULL };
enum { F_NEG = 1, F_ABS = 2, F_SEXT = 4 };
struct Opnd {
    u8 kind = K_NONE, flags = 0;
    u16 reg = 0, n = 1;      // first register, dwords
    i64 imm = 0;
    double f = 0.0;
};
enum Enc : u8 { E_PLAIN, E_E64, E_SDWA, E_DPP };
enum Sel : u8 { SEL_BYTE0, SEL_BYTE1, SEL_BYTE2, SEL_BYTE3, SEL_WORD0, SEL_WORD1, SEL_DWORD };
enum Unused : u8 { UNUSED_PAD, UNUSED_SEXT, UNUSED_PRESERVE };

struct Inst {
    u16 op = 0;
    u8 enc = E_PLAIN, no = 0;
    Opnd o[6];
    i32 off0 = 0, off1 = 0;
    u32 target = 0;              // index of the branch target
    u16 dpp = 0xFFFF;            // dpp_ctrl (0xFFFF: none)
    u8 row_mask = 0xF, bank_mask = 0xF;
    bool bound_ctrl = false, clamp = false, ret = false;   // ret (sc0): an atomic that returns the old value
    bool sc1 = false, nt = false;                           // cache policy bits of memory instructions (the L1 model looks at them)
    u8 dst_sel = SEL_DWORD, dst_unused = UNUSED_PAD, src0_sel = SEL_DWORD, src1_sel = SEL_DWORD;
    u8 bitop3 = 0;               // v_bitop3: the truth table
    u8 op_sel = 0, op_sel_hi = 7;
    u8 gpr_idx_mode = 0;         // s_set_gpr_idx_on: bit 0 SRC0, 1 SRC1, 2 SRC2, 3 DST
    u64 addr = 0;                // byte address inside the code object
    u32 line = 0;
};

struct KernArg { u32 offset, size; std::string kind; };
struct KernelInfo {
    std::string name;
    u64 entry = 0;               // address of the first instruction inside the code object
    u32 lds_static = 0, scratch = 0, kernarg_size = 0, vgprs = 0, agprs = 0, sgprs = 0;
    u32 rsrc1 = 0, rsrc2 = 0, rsrc3 = 0, props = 0;
    std::vector<KernArg> args;
    u32 first_inst = 0;
};
struct CodeObject {
    std::vector<u8> elf;         // the file
    std::vector<u8> image;       // loaded segments: image[vaddr]
    std::vector<Inst> insts;     // the whole .text, parsed on first use
    std::unordered_map<u64, u32> at;     // address -> index
    std::unordered_map<std::string, KernelInfo> kernels;
    bool parsed = false;
    std::string path;
};

const char *op_name(u16 op);
int op_lookup(const std::string &mnemonic, u8 *enc);
bool load_code_object(CodeObject &co, const u8 *elf, size_t size);
bool parse_text(CodeObject &co);

#define OPS(X) \
    X(s_mov_b32) X(s_mov_b64) X(s_movk_i32) X(s_and_b32) X(s_and_b64) X(s_or_b32) X(s_or_b64) X(s_xor_b32) X(s_xor_b64) \
    X(s_andn2_b32) X(s_andn2_b64) X(s_orn2_b32) X(s_orn2_b64) X(s_nor_b32) X(s_nor_b64) X(s_nand_b32) X(s_nand_b64) X(s_xnor_b32) X(s_xnor_b64) \
    X(s_not_b32) X(s_not_b64) \
    X(s_add_i32) X(s_add_u32) X(s_addc_u32) X(s_sub_i32) X(s_sub_u32) X(s_subb_u32) X(s_mul_i32) X(s_mul_hi_u32) X(s_mul_hi_i32) \
    X(s_lshl_b32) X(s_lshl_b64) X(s_lshr_b32) X(s_lshr_b64) X(s_ashr_i32) X(s_ashr_i64) X(s_bfe_u32) X(s_bfe_i32) X(s_bfe_u64) X(s_bfm_b32) X(s_bfm_b64) \
    X(s_min_u32) X(s_max_u32) X(s_min_i32) X(s_max_i32) X(s_cselect_b32) X(s_cselect_b64) X(s_abs_i32) X(s_sext_i32_i8) X(s_sext_i32_i16) \
    X(s_lshl1_add_u32) X(s_lshl2_add_u32) X(s_lshl3_add_u32) X(s_lshl4_add_u32) X(s_pack_ll_b32_b16) \
    X(s_and_saveexec_b64) X(s_or_saveexec_b64) X(s_andn2_saveexec_b64) X(s_xor_saveexec_b64) X(s_orn2_saveexec_b64) X(s_andn1_saveexec_b64) \
    X(s_brev_b32) X(s_brev_b64) X(s_bcnt1_i32_b32) X(s_bcnt1_i32_b64) X(s_bcnt0_i32_b32) X(s_ff1_i32_b32) X(s_ff1_i32_b64) X(s_ff0_i32_b32) X(s_flbit_i32_b32) X(s_flbit_i32_b64) \
    X(s_flbit_i32) X(s_bitset1_b32) X(s_bitset0_b32) X(s_bitset1_b64) X(s_bitset0_b64) X(s_bitcmp0_b32) X(s_bitcmp1_b32) X(s_bitcmp0_b64) X(s_bitcmp1_b64) \
    X(s_cmp_eq_i32) X(s_cmp_lg_i32) X(s_cmp_gt_i32) X(s_cmp_ge_i32) X(s_cmp_lt_i32) X(s_cmp_le_i32) \
    X(s_cmp_eq_u32) X(s_cmp_lg_u32) X(s_cmp_gt_u32) X(s_cmp_ge_u32) X(s_cmp_lt_u32) X(s_cmp_le_u32) X(s_cmp_eq_u64) X(s_cmp_lg_u64) \
    X(s_cmpk_eq_i32) X(s_cmpk_lg_i32) X(s_cmpk_gt_i32) X(s_cmpk_ge_i32) X(s_cmpk_lt_i32) X(s_cmpk_le_i32) \
    X(s_cmpk_eq_u32) X(s_cmpk_lg_u32) X(s_cmpk_gt_u32) X(s_cmpk_ge_u32) X(s_cmpk_lt_u32) X(s_cmpk_le_u32) \
    X(s_addk_i32) X(s_mulk_i32) X(s_getpc_b64) X(s_setpc_b64) X(s_swappc_b64) X(s_set_gpr_idx_on) X(s_set_gpr_idx_off) X(s_set_gpr_idx_idx) \
    X(s_branch) X(s_cbranch_scc0) X(s_cbranch_scc1) X(s_cbranch_vccz) X(s_cbranch_vccnz) X(s_cbranch_execz) X(s_cbranch_execnz) \
    X(s_endpgm) X(s_barrier) X(s_waitcnt) X(s_nop) X(s_sleep) X(s_memrealtime) X(s_memtime) X(s_trap) X(s_sethalt) X(s_setprio) X(s_code_end) \
    X(s_load_dword) X(s_load_dwordx2) X(s_load_dwordx4) X(s_load_dwordx8) X(s_load_dwordx16) X(s_dcache_wb) X(s_dcache_inv) X(s_icache_inv) \
    X(v_mov_b32) X(v_mov_b64) X(v_add_u32) X(v_sub_u32) X(v_subrev_u32) X(v_add_co_u32) X(v_addc_co_u32) X(v_sub_co_u32) X(v_subb_co_u32) \
    X(v_subrev_co_u32) X(v_subbrev_co_u32) X(v_mul_lo_u32) X(v_mul_hi_u32) X(v_mul_hi_i32) X(v_mul_i32_i24) X(v_mul_u32_u24) X(v_mul_hi_i32_i24) X(v_mul_hi_u32_u24) \
    X(v_mad_i32_i24) X(v_mad_u32_u24) X(v_mad_u64_u32) X(v_mad_i64_i32) X(v_and_b32) X(v_or_b32) X(v_xor_b32) X(v_xnor_b32) X(v_not_b32) \
    X(v_lshlrev_b32) X(v_lshrrev_b32) X(v_ashrrev_i32) X(v_lshlrev_b64) X(v_lshrrev_b64) X(v_ashrrev_i64) X(v_lshl_add_u32) X(v_lshl_add_u64) \
    X(v_add_lshl_u32) X(v_lshl_or_b32) X(v_and_or_b32) X(v_or3_b32) X(v_add3_u32) X(v_xad_u32) X(v_bfe_u32) X(v_bfe_i32) X(v_bfi_b32) X(v_bfm_b32) \
    X(v_bfrev_b32) X(v_alignbit_b32) X(v_alignbyte_b32) X(v_perm_b32) X(v_cndmask_b32) X(v_min_u32) X(v_max_u32) X(v_min_i32) X(v_max_i32) \
    X(v_min3_u32) X(v_max3_u32) X(v_med3_u32) X(v_min3_i32) X(v_max3_i32) X(v_med3_i32) X(v_sad_u32) X(v_ffbh_u32) X(v_ffbh_i32) X(v_ffbl_b32) X(v_bcnt_u32_b32) \
    X(v_mbcnt_lo_u32_b32) X(v_mbcnt_hi_u32_b32) X(v_readlane_b32) X(v_writelane_b32) X(v_readfirstlane_b32) X(v_bitop3_b32) X(v_bitop3_b16) \
    X(v_dot2_i32_i16) X(v_pk_mov_b32) X(v_accvgpr_read_b32) X(v_accvgpr_write_b32) X(v_accvgpr_mov_b32) X(v_lshlrev_b16) X(v_lshrrev_b16) X(v_ashrrev_i16) \
    X(v_add_u16) X(v_sub_u16) X(v_mul_lo_u16) X(v_max_u16) X(v_min_u16) X(v_max_i16) X(v_min_i16) X(v_mad_u16) X(v_cvt_f32_u32) X(v_cvt_f32_i32) X(v_cvt_u32_f32) X(v_cvt_i32_f32) \
    X(v_rcp_iflag_f32) X(v_rcp_f32) X(v_ldexp_f32) X(v_cvt_f32_ubyte0) X(v_cvt_f32_ubyte1) X(v_cvt_f32_ubyte2) X(v_cvt_f32_ubyte3) X(v_cvt_f32_f64) X(v_cvt_f64_f32) X(v_mul_f32) X(v_add_f32) X(v_sub_f32) X(v_subrev_f32) \
    X(v_fma_f32) X(v_fmac_f32) X(v_mac_f32) X(v_mad_f32) X(v_max_f32) X(v_min_f32) X(v_floor_f32) X(v_trunc_f32) X(v_rndne_f32) X(v_fract_f32) \
    X(v_add_f64) X(v_mul_f64) X(v_fma_f64) X(v_fmac_f64) X(v_max_f64) X(v_min_f64) X(v_floor_f64) X(v_trunc_f64) X(v_ceil_f64) X(v_rndne_f64) X(v_fract_f64) \
    X(v_cvt_f64_i32) X(v_cvt_f64_u32) X(v_cvt_i32_f64) X(v_cvt_u32_f64) X(v_ldexp_f64) X(v_frexp_mant_f64) X(v_frexp_exp_i32_f64) \
    X(v_div_scale_f64) X(v_div_fmas_f64) X(v_div_fixup_f64) X(v_rcp_f64) X(v_rsq_f64) X(v_sqrt_f64) X(v_cmp_class_f64) X(v_cmp_class_f32) \
    X(v_cmp_f_f64) X(v_cmp_lt_f64) X(v_cmp_eq_f64) X(v_cmp_le_f64) X(v_cmp_gt_f64) X(v_cmp_lg_f64) X(v_cmp_ge_f64) X(v_cmp_o_f64) X(v_cmp_u_f64) \
    X(v_cmp_nge_f64) X(v_cmp_nlg_f64) X(v_cmp_ngt_f64) X(v_cmp_nle_f64) X(v_cmp_neq_f64) X(v_cmp_nlt_f64) X(v_cmp_tru_f64) \
    X(v_cmp_f_f32) X(v_cmp_lt_f32) X(v_cmp_eq_f32) X(v_cmp_le_f32) X(v_cmp_gt_f32) X(v_cmp_lg_f32) X(v_cmp_ge_f32) X(v_cmp_o_f32) X(v_cmp_u_f32) \
    X(v_cmp_nge_f32) X(v_cmp_nlg_f32) X(v_cmp_ngt_f32) X(v_cmp_nle_f32) X(v_cmp_neq_f32) X(v_cmp_nlt_f32) X(v_cmp_tru_f32) \
    X(v_cmp_lt_u32) X(v_cmp_eq_u32) X(v_cmp_le_u32) X(v_cmp_gt_u32) X(v_cmp_ne_u32) X(v_cmp_ge_u32) \
    X(v_cmp_lt_i32) X(v_cmp_eq_i32) X(v_cmp_le_i32) X(v_cmp_gt_i32) X(v_cmp_ne_i32) X(v_cmp_ge_i32) \
    X(v_cmp_lt_u64) X(v_cmp_eq_u64) X(v_cmp_le_u64) X(v_cmp_gt_u64) X(v_cmp_ne_u64) X(v_cmp_ge_u64) \
    X(v_cmp_lt_i64) X(v_cmp_eq_i64) X(v_cmp_le_i64) X(v_cmp_gt_i64) X(v_cmp_ne_i64) X(v_cmp_ge_i64) \
    X(v_cmp_lt_u16) X(v_cmp_eq_u16) X(v_cmp_le_u16) X(v_cmp_gt_u16) X(v_cmp_ne_u16) X(v_cmp_ge_u16) \
    X(v_cmp_lt_i16) X(v_cmp_eq_i16) X(v_cmp_le_i16) X(v_cmp_gt_i16) X(v_cmp_ne_i16) X(v_cmp_ge_i16) \
    X(v_mfma_f64_4x4x4_4b_f64) X(v_nop) \
    X(ds_read_b32) X(ds_read_b64) X(ds_read_b96) X(ds_read_b128) X(ds_read_u8) X(ds_read_i8) X(ds_read_u16) X(ds_read_i16) X(ds_read_u16_d16) X(ds_read_u16_d16_hi) \
    X(ds_read2_b32) X(ds_read2_b64) X(ds_read2st64_b32) X(ds_read2st64_b64) \
    X(ds_write_b8) X(ds_write_b16) X(ds_write_b32) X(ds_write_b64) X(ds_write_b96) X(ds_write_b128) X(ds_write_b8_d16_hi) X(ds_write_b16_d16_hi) \
    X(ds_write2_b32) X(ds_write2_b64) X(ds_write2st64_b32) X(ds_write2st64_b64) \
    X(ds_or_b32) X(ds_xor_b32) X(ds_and_b32) X(ds_add_u32) X(ds_sub_u32) X(ds_max_u32) X(ds_min_u32) X(ds_max_i32) X(ds_min_i32) X(ds_inc_u32) \
    X(ds_add_rtn_u32) X(ds_or_rtn_b32) X(ds_max_rtn_u32) X(ds_min_rtn_u32) X(ds_inc_rtn_u32) X(ds_wrxchg_rtn_b32) X(ds_cmpst_rtn_b32) X(ds_add_u64) X(ds_or_b64) \
    X(ds_bpermute_b32) X(ds_permute_b32) X(ds_swizzle_b32) X(ds_nop) \
    X(global_load_ubyte) X(global_load_sbyte) X(global_load_ushort) X(global_load_sshort) X(global_load_dword) X(global_load_dwordx2) X(global_load_dwordx3) X(global_load_dwordx4) \
    X(global_load_short_d16) X(global_load_short_d16_hi) X(global_load_ubyte_d16) X(global_load_ubyte_d16_hi) \
    X(global_store_byte) X(global_store_short) X(global_store_dword) X(global_store_dwordx2) X(global_store_dwordx3) X(global_store_dwordx4) X(global_store_byte_d16_hi) X(global_store_short_d16_hi) \
    X(global_atomic_or) X(global_atomic_or_x2) X(global_atomic_and) X(global_atomic_and_x2) X(global_atomic_xor) X(global_atomic_add) X(global_atomic_add_x2) X(global_atomic_sub) X(global_atomic_sub_x2) \
    X(global_atomic_umin) X(global_atomic_umin_x2) X(global_atomic_umax) X(global_atomic_umax_x2) X(global_atomic_smin) X(global_atomic_smax) X(global_atomic_smin_x2) X(global_atomic_smax_x2) \
    X(global_atomic_inc) X(global_atomic_dec) X(global_atomic_swap) X(global_atomic_swap_x2) X(global_atomic_cmpswap) X(global_atomic_cmpswap_x2) \
    X(global_load_lds_dword) X(global_load_lds_dwordx3) X(global_load_lds_dwordx4) X(global_load_lds_ubyte) X(global_load_lds_ushort) \
    X(flat_load_ubyte) X(flat_load_sbyte) X(flat_load_ushort) X(flat_load_sshort) X(flat_load_dword) X(flat_load_dwordx2) X(flat_load_dwordx3) X(flat_load_dwordx4) \
    X(flat_store_byte) X(flat_store_short) X(flat_store_dword) X(flat_store_dwordx2) X(flat_store_dwordx3) X(flat_store_dwordx4) \
    X(flat_atomic_or) X(flat_atomic_or_x2) X(flat_atomic_add) X(flat_atomic_add_x2) X(flat_atomic_umax) X(flat_atomic_umin) X(flat_atomic_umax_x2) X(flat_atomic_umin_x2) X(flat_atomic_inc) X(flat_atomic_swap) X(flat_atomic_cmpswap) X(flat_atomic_cmpswap_x2) X(flat_atomic_and) \
    X(scratch_load_ubyte) X(scratch_load_sbyte) X(scratch_load_ushort) X(scratch_load_sshort) X(scratch_load_dword) X(scratch_load_dwordx2) X(scratch_load_dwordx3) X(scratch_load_dwordx4) \
    X(scratch_store_byte) X(scratch_store_short) X(scratch_store_dword) X(scratch_store_dwordx2) X(scratch_store_dwordx3) X(scratch_store_dwordx4) \
    X(buffer_inv) X(buffer_wbl2) X(buffer_wbinvl1) X(buffer_gl0_inv)

enum Op : u16 {
#define X(n) OP_##n,
    OPS(X)
#undef X
    OP_COUNT
};
