// fqsx_k_pe.hip -- the paired-end encode kernels (dna_mode 2: original order, 3: sorted by the first mate).
// One workgroup = one logical worker: resolve (both mates, minimizer anchoring: fqsx_pe.h), models, range coder,
// inserter and two scouts that serve one request per compress_suffix call (first mate after its prefix, second mate
// directly or right / left of its anchor); the paired-end scratch lies over the third scout's ring slot.
#include "fqsx_roles.h"

#ifndef FQSX_EMU
template <int MODE> FQ_DEV void encode_pe_kernel_body(const EncArgs &a) {
  if (worker_elsewhere(a)) return;
  if (wg_handoff_init(a, true)) return;   // a device error or a phase posted for growth stops the block's remaining launches (phase_skip)
  switch (FQ_WAVE_ID) {
    case 1: role_scout_req<2>(fq_kernarg(), 0u); break;
    case 2: role_resolve<MODE>(fq_kernarg()); break;
    case 3: role_models(fq_kernarg()); break;
    case 4: role_rc(fq_kernarg()); break;
    case 5: role_inserter(fq_kernarg()); break;
    case 7: role_scout_req<2>(fq_kernarg(), 1u); break;
    default: break;
  }
}
FQ_KERNEL512 void k_encode_pe_orig(EncArgs a) { encode_pe_kernel_body<2>(a); }
FQ_KERNEL512 void k_encode_pe_sorted(EncArgs a) { encode_pe_kernel_body<3>(a); }
int fqsx_launch_encode_pe(hipStream_t s, const EncArgs &a) {
  if (a.cfg.mode == 2) hipLaunchKernelGGL(k_encode_pe_orig, dim3(a.cfg.T), dim3(512), 0, s, a);
  else hipLaunchKernelGGL(k_encode_pe_sorted, dim3(a.cfg.T), dim3(512), 0, s, a);
  return (int)hipGetLastError();
}
#else
static void fqsx_emu_encode_pe(const EncArgs &a) {
  if (a.cfg.err[0] | a.cfg.err[1]) return;
  for (u32 b = 0; b < a.cfg.T; ++b) {
    fq_emu_block = b;
    if (!shard_mine(a.cfg, b)) {   // sharded run: a worker that lives on another rank only reports empty lists here
      for (u32 k = 0; k < 3; ++k) a.cfg.mail[k].n[b] = 0;
      a.cfg.pe_n[b] = 0;
      continue;
    }
    if (a.cfg.mode == 2) encode_segment_body<2, false, false>(a.cfg, fq_wg(), b, a.n_reads, a.S, a.seg);
    else encode_segment_body<3, false, false>(a.cfg, fq_wg(), b, a.n_reads, a.S, a.seg);
  }
}
#endif
