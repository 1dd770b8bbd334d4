// fqsx_roles.h -- the roles of a worker as functions of their own (FQ_ROLE), shared by the encode kernels
// (fqsx_k_se.hip, fqsx_k_pe.hip).  One workgroup = one logical worker; wave w runs on SIMD w % 4:
//   0 read head (single-end sorted only)   1, 7, 6 scouts   2 resolve   3 models   4 range coder   5 local-table inserter
// so the two busiest roles, resolve and models, share their SIMD only with a scout.
#pragma once
#include "fqsx_kernels.h"

template <int MODE> FQ_ROLE void role_resolve(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  encode_segment_body<MODE, false, true>(a->cfg, fq_wg(), FQ_BLOCK, a->n_reads, a->S, a->seg, a->pad);
}
FQ_ROLE void role_models(FqArgsP ap) {
  const EncArgs *a = fq_args(ap);
  coder_segment_body<true>(a->cfg, fq_wg(), FQ_BLOCK, a->seg, a->pad);
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
// The scout waves share ONE copy of their code (`me` is a run-time argument, made wave-uniform again inside): three
// instantiations were 3 x 38 KB of the kernel's ~390 KB of role code competing for the 64 KB instruction cache a CU
// pair shares, and three waves running the same lines fetch them once.
FQ_ROLE void role_scout(FqArgsP ap, u32 me_) {   // single-end sorted: the scouts walk the reads behind the read-head wave
  const EncArgs *a = fq_args(ap);
  scout_segment_body(a->cfg, fq_wg(), FQ_BLOCK, a->n_reads, a->S, a->seg, uniform32(me_), a->pad);
}
template <int NSC> FQ_ROLE void role_scout_req(FqArgsP ap, u32 me_) {   // the other modes: one request per compress_suffix call
  const EncArgs *a = fq_args(ap);
  scout_request_body(a->cfg, fq_wg(), FQ_BLOCK, uniform32(me_), (u32)NSC, a->pad);
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
// The one workgroup barrier of the kernel: the LDS hand-off words start at zero.  Returns true if the block's queue has
// been stopped (fqsx_api.hip: phase_skip) -- decided ONCE per workgroup (thread 0 reads the words, every wave branches on
// the LDS copy after the barrier), so that a word another workgroup of the same launch writes meanwhile cannot send
// some waves of a worker home and the rest into the hand-off protocol without their partners.
FQ_DEV bool wg_handoff_init(const EncArgs &a, bool honour_posted) {
  WgShared *sm = fq_wg();
  if (threadIdx.x == 0) {
    sm->wg_stop = (a.cfg.err[0] | (honour_posted ? a.cfg.err[1] : 0u)) != 0 ? 1u : 0u;
    sm->cq_tail = 0; sm->cq_head = 0; sm->cq_done = 0;
    sm->lq_target[0] = sm->lq_target[1] = 0; sm->lq_done[0] = sm->lq_done[1] = 0; sm->lq_quit = 0;
    sm->hd_ready = 0; sm->hd_taken = 0; sm->hd_early = 0;
    sm->sc_taken = 0; sm->sc_req_seq = 0; sm->sc_dead = (a.cfg.dbg & FQSX_DBG_SCOUTS_OFF) ? 1u : 0u;
    for (u32 x = 0; x < FQSX_NSC; ++x) { sm->sc_hd_taken[x] = 0; sm->sc_ack[x] = 0; }
    for (u32 x = 1; x <= FQSX_SCR; ++x) sm->sb[x].h_pub = 0;
    sm->rq_tail = 0; sm->rq_head = 0; sm->rq_done = 0;
  }
  // partitioned tables: other GPUs' sub-tables have changed since this CU / XCD last cached lines of them (one wave's
  // acquire serves the workgroup: every wave's loads go through the same L1 and L2, and all of them wait at the barrier)
  if (a.cfg.sys_scope && FQ_WAVE_ID == 0) fq_acquire_system();
  FQ_WG_BARRIER();
  return sm->wg_stop != 0;
}
#endif
