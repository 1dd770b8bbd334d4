// fqsx_k_dec.hip -- the decode kernels (fqsx_dec.h): one wavefront per worker, every position through the complete
// logic (the next k-mer depends on the symbol being decoded, so there is no stage P).
#include "fqsx_kernels.h"

// (one wave per workgroup: no roles to separate, the body is inlined into the kernel, which gets the whole register file)
template <int MODE> FQ_DEV void decode_kernel_body(const EncArgs &a) {
  if (a.cfg.err[0]) return;   // (one wave per workgroup: the test is the workgroup's) a device error stops the block's remaining launches
  if (a.cfg.sys_scope) fq_acquire_system();   // partitioned tables: see wg_handoff_init
  encode_segment_body<MODE, true, false>(a.cfg, fq_wg(), FQ_BLOCK, a.n_reads, a.S, a.seg, a.pad);
}

#ifndef FQSX_EMU
FQ_KERNEL64 void k_decode_se_orig(EncArgs a) { decode_kernel_body<0>(a); }
FQ_KERNEL64 void k_decode_se_sorted(EncArgs a) { decode_kernel_body<1>(a); }
FQ_KERNEL64 void k_decode_pe_orig(EncArgs a) { decode_kernel_body<2>(a); }
FQ_KERNEL64 void k_decode_pe_sorted(EncArgs a) { decode_kernel_body<3>(a); }
int fqsx_launch_decode(hipStream_t s, const EncArgs &a) {
  const dim3 g(a.cfg.T), b(64);
  switch (a.cfg.mode) {
    case 0: hipLaunchKernelGGL(k_decode_se_orig, g, b, 0, s, a); break;
    case 1: hipLaunchKernelGGL(k_decode_se_sorted, g, b, 0, s, a); break;
    case 2: hipLaunchKernelGGL(k_decode_pe_orig, g, b, 0, s, a); break;
    default: hipLaunchKernelGGL(k_decode_pe_sorted, g, b, 0, s, a); break;
  }
  return (int)hipGetLastError();
}
#else
static void fqsx_emu_decode(const EncArgs &a) {
    for (u32 b = 0; b < a.cfg.T; ++b) {
    fq_emu_block = b;
    switch (a.cfg.mode) {
      case 0: encode_segment_body<0, true, false>(a.cfg, fq_wg(), b, a.n_reads, a.S, a.seg); break;
      case 1: encode_segment_body<1, true, false>(a.cfg, fq_wg(), b, a.n_reads, a.S, a.seg); break;
      case 2: encode_segment_body<2, true, false>(a.cfg, fq_wg(), b, a.n_reads, a.S, a.seg); break;
      default: encode_segment_body<3, true, false>(a.cfg, fq_wg(), b, a.n_reads, a.S, a.seg); break;
    }
  }
}
#endif
