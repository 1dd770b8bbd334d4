// fqsx_kernels.h -- launchers of the encode / decode kernels, one translation unit per kernel family
// (fqsx_k_se.hip, fqsx_k_pe.hip, fqsx_k_dec.hip) so that they compile in parallel and every kernel holds only the
// code of its own dna_mode.  The host side (fqsx_api.hip) calls these; the FQSX_EMU build includes the same files
// into one translation unit and runs the kernels as plain loops.
#pragma once
#include "fqsx_dev.h"

#ifndef FQSX_EMU
// Each returns the hipError_t of the launch as an int (0 = hipSuccess).
int fqsx_launch_encode_se(hipStream_t s, const EncArgs &a);   // dna_mode 0 (original order) and 1 (sorted)
int fqsx_launch_encode_pe(hipStream_t s, const EncArgs &a);   // dna_mode 2 and 3
int fqsx_launch_decode(hipStream_t s, const EncArgs &a);      // all modes
#else
static void fqsx_emu_encode_se(const EncArgs &a);
static void fqsx_emu_encode_pe(const EncArgs &a);
static void fqsx_emu_decode(const EncArgs &a);
#endif
