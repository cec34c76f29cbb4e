// glue.hip -- the single-sweep kernels that sit between advection and the pressure solve in every smoke /
// FLIP scene, so that a step never leaves the device.  Reference: source/plugin/extforces.cpp.
#include "common.h"

using namespace mf;

#define CELL_IJK(d)                                                               \
	const int64_t idx = blockIdx.x * (int64_t)BLOCK + threadIdx.x;                \
	if (idx >= (d).n) return;                                                     \
	const int i = (int)(idx % (d).sx);                                            \
	const int j = (int)((idx / (d).sx) % (d).sy);                                 \
	const int k = (int)(idx / ((int64_t)(d).sx * (d).sy));                        \
	(void)i; (void)j; (void)k;
#define INTERIOR(d) (i >= 1 && i < (d).sx - 1 && j >= 1 && j < (d).sy - 1 && (!(d).is3d || (k >= 1 && k < (d).sz - 1)))
static inline unsigned nblk(const Dim& d) { return (unsigned)((d.n + BLOCK - 1) / BLOCK); }

// KnSetWallBcs, extforces.cpp:187-237 (KERNEL(): every cell; each thread writes only its own cell)
__global__ void __launch_bounds__(BLOCK)
k_set_wall_bcs(Dim d, const int32_t* __restrict__ flags, float* __restrict__ vel, const float* __restrict__ obvel) {
	CELL_IJK(d)
	const int64_t n = d.n, Y = d.Y, Z = d.Z;
	const int f = flags[idx];
	const bool curFluid = f & MF_FLUID, curObs = f & MF_OBSTACLE;
	if (!curFluid && !curObs) return;
	float bx = 0.f, by = 0.f, bz = 0.f;
	if (obvel) {
		bx = obvel[idx];
		by = obvel[n + idx];
		if (d.is3d) bz = obvel[2 * n + idx];
	}
	float vx = vel[idx], vy = vel[n + idx], vz = vel[2 * n + idx];
	const int fxm = i > 0 ? flags[idx - 1] : 0, fym = j > 0 ? flags[idx - Y] : 0, fzm = (d.is3d && k > 0) ? flags[idx - Z] : 0;
	if (i > 0 && (fxm & MF_OBSTACLE)) vx = bx;
	if (i > 0 && curObs && (fxm & MF_FLUID)) vx = bx;
	if (j > 0 && (fym & MF_OBSTACLE)) vy = by;
	if (j > 0 && curObs && (fym & MF_FLUID)) vy = by;
	if (!d.is3d) {
		vz = 0.f;
	} else {
		if (k > 0 && (fzm & MF_OBSTACLE)) vz = bz;
		if (k > 0 && curObs && (fzm & MF_FLUID)) vz = bz;
	}
	if (curFluid) {
		if ((i > 0 && (fxm & MF_STICK)) || (i < d.sx - 1 && (flags[idx + 1] & MF_STICK))) vy = vz = 0.f;
		if ((j > 0 && (fym & MF_STICK)) || (j < d.sy - 1 && (flags[idx + Y] & MF_STICK))) vx = vz = 0.f;
		if (d.is3d && ((k > 0 && (fzm & MF_STICK)) || (k < d.sz - 1 && (flags[idx + Z] & MF_STICK)))) vx = vy = 0.f;
	}
	vel[idx] = vx;
	vel[n + idx] = vy;
	vel[2 * n + idx] = vz;
}

// KnAddBuoyancy, extforces.cpp:73-81: `vel += (0.5*strength) * (f0 + f1)` is an fp64 compound assignment
__global__ void __launch_bounds__(BLOCK)
k_add_buoyancy(Dim d, const int32_t* __restrict__ flags, const float* __restrict__ fac, float* __restrict__ vel, float fx, float fy, float fz) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	if (!(flags[idx] & MF_FLUID)) return;
	const int64_t n = d.n, Y = d.Y, Z = d.Z;
	const float f0 = fac[idx];
	if (flags[idx - 1] & MF_FLUID) vel[idx] = (float)((double)vel[idx] + (0.5 * (double)fx) * (double)(f0 + fac[idx - 1]));
	if (flags[idx - Y] & MF_FLUID) vel[n + idx] = (float)((double)vel[n + idx] + (0.5 * (double)fy) * (double)(f0 + fac[idx - Y]));
	if (d.is3d && (flags[idx - Z] & MF_FLUID))
		vel[2 * n + idx] = (float)((double)vel[2 * n + idx] + (0.5 * (double)fz) * (double)(f0 + fac[idx - Z]));
}

// KnApplyForce, extforces.cpp:46-60
__global__ void __launch_bounds__(BLOCK)
k_apply_force(Dim d, const int32_t* __restrict__ flags, float* __restrict__ vel, float fx, float fy, float fz,
              const float* __restrict__ exclude, int additive) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	const int64_t n = d.n, Y = d.Y, Z = d.Z;
	const bool curFluid = flags[idx] & MF_FLUID, curEmpty = flags[idx] & MF_EMPTY;
	if (!curFluid && !curEmpty) return;
	if (exclude && (exclude[idx] < 0.f)) return;
	const int fxm = flags[idx - 1], fym = flags[idx - Y];
	if ((fxm & MF_FLUID) || (curFluid && (fxm & MF_EMPTY))) vel[idx] = additive ? vel[idx] + fx : fx;
	if ((fym & MF_FLUID) || (curFluid && (fym & MF_EMPTY))) vel[n + idx] = additive ? vel[n + idx] + fy : fy;
	if (d.is3d) {
		const int fzm = flags[idx - Z];
		if ((fzm & MF_FLUID) || (curFluid && (fzm & MF_EMPTY))) vel[2 * n + idx] = additive ? vel[2 * n + idx] + fz : fz;
	}
}

extern "C" {
int mf_set_wall_bcs(int sx, int sy, int sz, const int32_t* flags, float* vel, const float* obvel, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_set_wall_bcs, dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, vel, obvel);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_add_buoyancy(int sx, int sy, int sz, const int32_t* flags, const float* density, float* vel, float fx, float fy, float fz, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_add_buoyancy, dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, density, vel, fx, fy, fz);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_apply_force(int sx, int sy, int sz, const int32_t* flags, float* vel, float fx, float fy, float fz, const float* exclude,
                   int additive, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_apply_force, dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, vel, fx, fy, fz, exclude, additive);
	MF_LAUNCH_CHECK();
	return 0;
}
}  // extern "C"
