// fqsx_k_se.hip -- the single-end encode kernels.
//
// One workgroup = one logical worker.  Sorted order (dna_mode 1, the benchmark's mode): eight wavefronts with fixed
// roles -- read head (duplicate test, p-mer prefix of the next read), three scouts (stage P: k-mer rolling and table
// probes, one position per lane, of the chunks ahead, taking the chunks in turn), resolve (k-mer tables, counts,
// corrections, mailboxes; queues every symbol in LDS), models (context search, model statistics, level averages), range
// coder (the sequential coding step and the output bytes), inserter (the worker's local-table inserts).
// Original order (dna_mode 0): the same without the read-head wave; the scouts serve one request per read.
// Every role is a function of its own (fqsx_roles.h): own register allocation, own stretch of code.
#include "fqsx_roles.h"

#ifndef FQSX_EMU
FQ_KERNEL512 void k_encode_se_sorted(EncArgs a) {
  if (worker_elsewhere(a)) return;
  if (wg_handoff_init(a, true)) return;   // the block's queue has been stopped (fqsx_api.hip: phase_skip)
  switch (FQ_WAVE_ID) {
    case 0: role_head(fq_kernarg()); break;
    case 1: role_scout(fq_kernarg(), 0u); break;
    case 2: role_resolve<1>(fq_kernarg()); break;
    case 3: role_models(fq_kernarg()); break;
    case 4: role_rc(fq_kernarg()); break;
    case 5: role_inserter(fq_kernarg()); break;
    case 6: role_scout(fq_kernarg(), 2u); break;
    default: role_scout(fq_kernarg(), 1u); break;
  }
}
FQ_KERNEL512 void k_encode_se_orig(EncArgs a) {
  if (worker_elsewhere(a)) return;
  if (wg_handoff_init(a, true)) return;
  switch (FQ_WAVE_ID) {
    case 0: break;
    case 1: role_scout_req<3>(fq_kernarg(), 0u); break;
    case 2: role_resolve<0>(fq_kernarg()); break;
    case 3: role_models(fq_kernarg()); break;
    case 4: role_rc(fq_kernarg()); break;
    case 5: role_inserter(fq_kernarg()); break;
    case 6: role_scout_req<3>(fq_kernarg(), 2u); break;
    default: role_scout_req<3>(fq_kernarg(), 1u); break;
  }
}
int fqsx_launch_encode_se(hipStream_t s, const EncArgs &a) {
  if (a.cfg.mode == 1) hipLaunchKernelGGL(k_encode_se_sorted, dim3(a.cfg.T), dim3(512), 0, s, a);
  else hipLaunchKernelGGL(k_encode_se_orig, dim3(a.cfg.T), dim3(512), 0, s, a);
  return (int)hipGetLastError();
}
#else
// host emulation: one 1-lane "wave" per worker runs the resolving body with everything inline
static void fqsx_emu_encode_se(const EncArgs &a) {
  if (a.cfg.err[0] | a.cfg.err[1]) return;
  for (u32 b = 0; b < a.cfg.T; ++b) {
    fq_emu_block = b;
    if (!shard_mine(a.cfg, b)) { for (u32 k = 0; k < 3; ++k) a.cfg.mail[k].n[b] = 0; continue; }
    if (a.cfg.mode == 1) encode_segment_body<1, false, false>(a.cfg, fq_wg(), b, a.n_reads, a.S, a.seg);
    else encode_segment_body<0, false, false>(a.cfg, fq_wg(), b, a.n_reads, a.S, a.seg);
  }
}
#endif
