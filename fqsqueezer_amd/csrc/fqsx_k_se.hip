// fqsx_k_se.hip -- the single-end encode kernels.
//
// One workgroup = one logical worker.  Sorted order (dna_mode 1, the benchmark's mode): five wavefronts with fixed
// roles -- wave 0 prepares the head of the next read (duplicate test, p-mer prefix), wave 1 resolves the worker's reads
// (k-mer tables, counts, corrections, mailboxes) and queues every symbol in LDS, wave 2 codes them (context models +
// range coder), wave 3 runs stage P (k-mer rolling and table probes, one position per lane) of the chunks ahead,
// wave 4 applies the worker's local-table inserts.  Original order (dna_mode 0): resolve, coder, inserter.
// Every role is a function of its own (FQ_ROLE): own register allocation, own stretch of code.
#include "fqsx_kernels.h"

template <int MODE> FQ_ROLE void role_resolve(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  encode_segment_body<MODE, false, true>(a->cfg, fq_wg(), FQ_BLOCK, a->n_reads, a->S, a->seg, a->pad);
}
FQ_ROLE void role_coder(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  coder_segment_body(a->cfg, fq_wg(), FQ_BLOCK, a->seg, a->pad);
}
FQ_ROLE void role_inserter(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  inserter_segment_body(a->cfg, fq_wg(), FQ_BLOCK, a->pad);
}
FQ_ROLE void role_head(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  head_segment_body(a->cfg, fq_wg(), FQ_BLOCK, a->n_reads, a->S, a->seg, a->pad);
}
FQ_ROLE void role_scout(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  scout_segment_body(a->cfg, fq_wg(), FQ_BLOCK, a->n_reads, a->S, a->seg, a->pad);
}

#ifndef FQSX_EMU
FQ_DEV void wg_handoff_init() {   // the one workgroup barrier of the kernel: the LDS hand-off words start at zero
  WgShared *sm = fq_wg();
  if (threadIdx.x == 0) {
    sm->cq_tail = 0; sm->cq_head = 0; sm->cq_done = 0;
    sm->lq_target[0] = sm->lq_target[1] = 0; sm->lq_done[0] = sm->lq_done[1] = 0; sm->lq_quit = 0;
    sm->hd_ready = 0; sm->hd_taken = 0;
    sm->sc_ready = 0; sm->sc_taken = 0; sm->sc_skip = 0; sm->sc_hd_taken = 0;
  }
  FQ_WG_BARRIER();
}
FQ_KERNEL320 void k_encode_se_sorted(EncArgs a) {
  (void)a;
  wg_handoff_init();
  // (waves 0 and 4 land on the same SIMD: the two least busy roles share it)
  if (FQ_WAVE_ID == 1) role_resolve<1>(fq_kernarg());
  else if (FQ_WAVE_ID == 2) role_coder(fq_kernarg());
  else if (FQ_WAVE_ID == 4) role_inserter(fq_kernarg());
  else if (FQ_WAVE_ID == 0) role_head(fq_kernarg());
  else role_scout(fq_kernarg());
}
FQ_KERNEL192 void k_encode_se_orig(EncArgs a) {
  (void)a;
  wg_handoff_init();
  if (FQ_WAVE_ID == 0) role_resolve<0>(fq_kernarg());
  else if (FQ_WAVE_ID == 1) role_coder(fq_kernarg());
  else role_inserter(fq_kernarg());
}
int fqsx_launch_encode_se(hipStream_t s, const EncArgs &a) {
  if (a.cfg.mode == 1) hipLaunchKernelGGL(k_encode_se_sorted, dim3(a.cfg.T), dim3(320), 0, s, a);
  else hipLaunchKernelGGL(k_encode_se_orig, dim3(a.cfg.T), dim3(192), 0, s, a);
  return (int)hipGetLastError();
}
#else
// host emulation: one 1-lane "wave" per worker runs the resolving body with everything inline
static void fqsx_emu_encode_se(const EncArgs &a) {
    for (u32 b = 0; b < a.cfg.T; ++b) {
    fq_emu_block = b;
    if (a.cfg.mode == 1) encode_segment_body<1, false, false>(a.cfg, fq_wg(), b, a.n_reads, a.S, a.seg);
    else encode_segment_body<0, false, false>(a.cfg, fq_wg(), b, a.n_reads, a.S, a.seg);
  }
}
#endif
