// Launcher interface of the big-D MFMA chain kernel (c3p_bigd.hip).
#pragma once
#include "c3p_common.h"
#include "c3p_midd.h"

#define C3P_BIGD_MAX_WGS 256  // one workgroup per CU; each owns a 7-image global arena

bool c3p_bigd_geometry(int Dm, int* nig, int* nj, int* w);
size_t c3p_bigd_table_doubles(int Dm, int K);
size_t c3p_bigd_arena_doubles(int Dm);
size_t c3p_bigd_lds_bytes(int Dm, int K, int Lmax);
hipError_t c3p_launch_bigd_chain(const MidArgs& A, double* arena, hipStream_t st);
hipError_t c3p_launch_rowphase(cplx* U, const double* phase, int B, int Dm, hipStream_t st);
