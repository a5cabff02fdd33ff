/* TEST infrastructure (never shipped): the compute entry points of include/mi355x_kernels.h for the ThreadSanitizer run of the plugin's host
 * logic — every one succeeds without doing anything: ONE zero-returning function under all their names (the x86-64 calling convention lets a
 * callee ignore its arguments).  Plain C without the header, so that the aliases do not collide with the real prototypes. */
__attribute__((visibility("default"))) int mi355x_stub_ok(void) { return 0; }
#define OK(name) __attribute__((visibility("default"), alias("mi355x_stub_ok"))) int name(void);
OK(mi355x_act_prepare) OK(mi355x_argmax_top2) OK(mi355x_binary) OK(mi355x_checksum) OK(mi355x_concat) OK(mi355x_cpy) OK(mi355x_ctx_synchronize)
OK(mi355x_decode_head_multi) OK(mi355x_dequant_f16) OK(mi355x_flash_attn_combine) OK(mi355x_flash_attn_ext) OK(mi355x_flash_attn_ext_exact)
OK(mi355x_flash_attn_ext_prep) OK(mi355x_flash_attn_ext_prep_rows) OK(mi355x_flash_attn_planes) OK(mi355x_flush) OK(mi355x_gelu) OK(mi355x_gemm_f16act)
OK(mi355x_gemm_f16act_prep) OK(mi355x_gemm_q8act) OK(mi355x_gemm_q8act_prep) OK(mi355x_gemv_fused) OK(mi355x_get_rows) OK(mi355x_get_rows_add)
OK(mi355x_im2col_1d) OK(mi355x_mul_mat) OK(mi355x_norm) OK(mi355x_norm_prep) OK(mi355x_pad_reflect_1d) OK(mi355x_prep_act) OK(mi355x_rope)
OK(mi355x_scale) OK(mi355x_scatter_upload) OK(mi355x_soft_max) OK(mi355x_unary) OK(mi355x_memset) OK(mi355x_wake) OK(mi355x_log_mel)
