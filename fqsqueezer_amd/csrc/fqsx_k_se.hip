// fqsx_k_se.hip -- the single-end encode kernels.
//
// One workgroup = one logical worker.  Sorted order (dna_mode 1, the benchmark's mode): seven wavefronts with fixed
// roles -- read head (duplicate test, p-mer prefix of the next read), two scouts (stage P: k-mer rolling and table probes,
// one position per lane, of the chunks ahead, taking the chunks in turn), resolve (k-mer tables, counts, corrections, mailboxes; queues every
// symbol in LDS), models (context search, model statistics, level averages), range coder (the sequential coding step
// and the output bytes), inserter (the worker's local-table inserts).  The two busiest roles, resolve and models, have
// a SIMD to themselves (waves 2 and 3); head + range coder share SIMD 0, scout + inserter SIMD 1.
// Original order (dna_mode 0): resolve, coder (models + range coder), inserter.
// Every role is a function of its own (FQ_ROLE): own register allocation, own stretch of code.
#include "fqsx_kernels.h"

template <int MODE> FQ_ROLE void role_resolve(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  encode_segment_body<MODE, false, true>(a->cfg, fq_wg(), FQ_BLOCK, a->n_reads, a->S, a->seg, a->pad);
}
template <bool SPLIT> FQ_ROLE void role_coder(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  coder_segment_body<SPLIT>(a->cfg, fq_wg(), FQ_BLOCK, a->seg, a->pad);
}
FQ_ROLE void role_rc(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  rc_segment_body(a->cfg, fq_wg(), FQ_BLOCK, a->seg, a->pad);
}
FQ_ROLE void role_inserter(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  inserter_segment_body(a->cfg, fq_wg(), FQ_BLOCK, a->pad);
}
FQ_ROLE void role_head(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  head_segment_body(a->cfg, fq_wg(), FQ_BLOCK, a->n_reads, a->S, a->seg, a->pad);
}
template <int ME> FQ_ROLE void role_scout(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  scout_segment_body(a->cfg, fq_wg(), FQ_BLOCK, a->n_reads, a->S, a->seg, (u32)ME, a->pad);
}

#ifndef FQSX_EMU
// sharded run: a worker that lives on another GPU only reports empty mailboxes here
FQ_DEV bool worker_elsewhere(const EncArgs &a) {
  if (shard_mine(a.cfg, FQ_BLOCK)) return false;
  if (threadIdx.x == 0) {
    for (u32 k = 0; k < 3; ++k) a.cfg.mail[k].n[FQ_BLOCK] = 0;
    if (a.cfg.pe_n) a.cfg.pe_n[FQ_BLOCK] = 0;
  }
  return true;
}
FQ_DEV void wg_handoff_init() {   // the one workgroup barrier of the kernel: the LDS hand-off words start at zero
  WgShared *sm = fq_wg();
  if (threadIdx.x == 0) {
    sm->cq_tail = 0; sm->cq_head = 0; sm->cq_done = 0;
    sm->lq_target[0] = sm->lq_target[1] = 0; sm->lq_done[0] = sm->lq_done[1] = 0; sm->lq_quit = 0;
    sm->hd_ready = 0; sm->hd_taken = 0;
    sm->sc_taken = 0; sm->sc_req_seq = 0; sm->sc_dead = 0;
    for (u32 x = 0; x < FQSX_NSC; ++x) { sm->sc_hd_taken[x] = 0; sm->sc_ack[x] = 0; }
    for (u32 x = 1; x <= FQSX_SCR; ++x) sm->sb[x].h_pub = 0;
    sm->rq_tail = 0; sm->rq_head = 0; sm->rq_done = 0;
  }
  FQ_WG_BARRIER();
}
FQ_KERNEL512 void k_encode_se_sorted(EncArgs a) {
  if (worker_elsewhere(a)) return;
  wg_handoff_init();
  // wave w runs on SIMD w % 4: resolve and models (+ the second scout) are the busy ones
  switch (FQ_WAVE_ID) {
    case 0: role_head(fq_kernarg()); break;
    case 1: role_scout<0>(fq_kernarg()); break;
    case 2: role_resolve<1>(fq_kernarg()); break;
    case 3: role_coder<true>(fq_kernarg()); break;
    case 4: role_rc(fq_kernarg()); break;
    case 5: role_inserter(fq_kernarg()); break;
    case 6: role_scout<2>(fq_kernarg()); break;
    default: role_scout<1>(fq_kernarg()); break;
  }
}
FQ_KERNEL192 void k_encode_se_orig(EncArgs a) {
  if (worker_elsewhere(a)) return;
  wg_handoff_init();
  if (FQ_WAVE_ID == 0) role_resolve<0>(fq_kernarg());
  else if (FQ_WAVE_ID == 1) role_coder<false>(fq_kernarg());
  else role_inserter(fq_kernarg());
}
int fqsx_launch_encode_se(hipStream_t s, const EncArgs &a) {
  if (a.cfg.mode == 1) hipLaunchKernelGGL(k_encode_se_sorted, dim3(a.cfg.T), dim3(512), 0, s, a);
  else hipLaunchKernelGGL(k_encode_se_orig, dim3(a.cfg.T), dim3(192), 0, s, a);
  return (int)hipGetLastError();
}
#else
// host emulation: one 1-lane "wave" per worker runs the resolving body with everything inline
static void fqsx_emu_encode_se(const EncArgs &a) {
    for (u32 b = 0; b < a.cfg.T; ++b) {
    fq_emu_block = b;
    if (!shard_mine(a.cfg, b)) { for (u32 k = 0; k < 3; ++k) a.cfg.mail[k].n[b] = 0; continue; }
    if (a.cfg.mode == 1) encode_segment_body<1, false, false>(a.cfg, fq_wg(), b, a.n_reads, a.S, a.seg);
    else encode_segment_body<0, false, false>(a.cfg, fq_wg(), b, a.n_reads, a.S, a.seg);
  }
}
#endif
