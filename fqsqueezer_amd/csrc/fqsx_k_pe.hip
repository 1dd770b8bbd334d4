// fqsx_k_pe.hip -- the paired-end encode kernels (dna_mode 2: original order, 3: sorted by the first mate).
// One workgroup = one logical worker = three wavefronts: resolve (both mates, minimizer anchoring: fqsx_pe.h),
// coder, local-table inserter.
#include "fqsx_kernels.h"

template <int MODE> FQ_ROLE void role_resolve_pe(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  encode_segment_body<MODE, false, true>(a->cfg, fq_wg(), FQ_BLOCK, a->n_reads, a->S, a->seg, a->pad);
}
FQ_ROLE void role_coder_pe(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  coder_segment_body<false>(a->cfg, fq_wg(), FQ_BLOCK, a->seg, a->pad);
}
FQ_ROLE void role_inserter_pe(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  inserter_segment_body(a->cfg, fq_wg(), FQ_BLOCK, a->pad);
}

#ifndef FQSX_EMU
template <int MODE> FQ_DEV void encode_pe_kernel_body() {
  WgShared *sm = fq_wg();
  if (threadIdx.x == 0) {
    sm->cq_tail = 0; sm->cq_head = 0; sm->cq_done = 0;
    sm->lq_target[0] = sm->lq_target[1] = 0; sm->lq_done[0] = sm->lq_done[1] = 0; sm->lq_quit = 0;
    sm->hd_ready = 0; sm->hd_taken = 0;
    sm->sc_taken = 0; sm->sc_req_seq = 0; sm->sc_dead = 0;
  }
  FQ_WG_BARRIER();
  if (FQ_WAVE_ID == 0) role_resolve_pe<MODE>(fq_kernarg());
  else if (FQ_WAVE_ID == 1) role_coder_pe(fq_kernarg());
  else role_inserter_pe(fq_kernarg());
}
FQ_KERNEL192 void k_encode_pe_orig(EncArgs a) { (void)a; encode_pe_kernel_body<2>(); }
FQ_KERNEL192 void k_encode_pe_sorted(EncArgs a) { (void)a; encode_pe_kernel_body<3>(); }
int fqsx_launch_encode_pe(hipStream_t s, const EncArgs &a) {
  if (a.cfg.mode == 2) hipLaunchKernelGGL(k_encode_pe_orig, dim3(a.cfg.T), dim3(192), 0, s, a);
  else hipLaunchKernelGGL(k_encode_pe_sorted, dim3(a.cfg.T), dim3(192), 0, s, a);
  return (int)hipGetLastError();
}
#else
static void fqsx_emu_encode_pe(const EncArgs &a) {
    for (u32 b = 0; b < a.cfg.T; ++b) {
    fq_emu_block = b;
    if (a.cfg.mode == 2) encode_segment_body<2, false, false>(a.cfg, fq_wg(), b, a.n_reads, a.S, a.seg);
    else encode_segment_body<3, false, false>(a.cfg, fq_wg(), b, a.n_reads, a.S, a.seg);
  }
}
#endif
