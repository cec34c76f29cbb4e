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

// ---- FLIP glue ---------------------------------------------------------------------------------
// marking pass of extrapolateMACSimple (serial FOR_IJK_BND in the reference, fastmarch.cpp:344-356): tmp = 1 where the
// face belongs to the fluid
__global__ void __launch_bounds__(BLOCK)
k_extrap_mark(Dim d, const int32_t* __restrict__ flags, int32_t* __restrict__ tmp, int c, int intoObs) {
	CELL_IJK(d)
	int v = 0;
	if (INTERIOR(d)) {
		const int64_t o = c == 0 ? 1 : (c == 1 ? d.Y : d.Z);
		const int f0 = flags[idx], f1 = flags[idx - o];
		bool mark = (f0 & MF_FLUID) || (f1 & MF_FLUID);
		if (intoObs) mark = mark && !(f0 & MF_OBSTACLE) && !(f1 & MF_OBSTACLE);
		v = mark ? 1 : 0;
	}
	tmp[idx] = v;
}
// knExtrapolateMACSimple, fastmarch.cpp:231-259.  In place like the reference: a pass only reads cells whose marker
// equals d and only writes cells whose marker is 0 (to d+1), so the result does not depend on the thread order.
__global__ void __launch_bounds__(BLOCK)
k_extrap_simple(Dim d, float* __restrict__ velc, int32_t* __restrict__ tmp, int dd) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	if (tmp[idx] != 0) return;
	const int64_t nb[6] = {1, -1, d.Y, -d.Y, d.Z, -d.Z};
	const int nn = d.is3d ? 6 : 4;
	int nbs = 0;
	float avg = 0.f;
	for (int n = 0; n < nn; n++)
		if (tmp[idx + nb[n]] == dd) {
			avg += velc[idx + nb[n]];
			nbs++;
		}
	if (nbs > 0) {
		tmp[idx] = dd + 1;
		velc[idx] = avg / (float)nbs;
	}
}
// The same pass with four consecutive cells per thread (3-D): the markers of the quad and of its +-Y / +-Z neighbour quads come as
// 16-byte loads (the +-Y ones from 4-byte aligned addresses when the row length is not a multiple of 4), 7 load instructions per 4 cells
// instead of 28 -- nearly every cell of a liquid scene only ever reads markers (12 passes per extrapolateMACSimple call).  Neighbour
// order and arithmetic per cell as above.
struct __attribute__((packed, aligned(4))) I4u {
	int x, y, z, w;
};
__global__ void __launch_bounds__(BLOCK)
k_extrap_simple4(Dim d, float* __restrict__ velc, int32_t* __restrict__ tmp, int dd) {
	const int64_t idx0 = 4 * (blockIdx.x * (int64_t)BLOCK + threadIdx.x);
	if (idx0 + 3 >= d.n) {
		// (the last, partial quad of a grid whose cell count is not a multiple of 4: border cells of the last plane -- never interior)
		return;
	}
	const int4 m4 = *(const int4*)(tmp + idx0);
	const int m[4] = {m4.x, m4.y, m4.z, m4.w};
	if (m[0] != 0 && m[1] != 0 && m[2] != 0 && m[3] != 0) return;
	const int64_t Y = d.Y, Z = d.Z;
	int i = (int)(idx0 % d.sx), j = (int)((idx0 / d.sx) % d.sy), k = (int)(idx0 / ((int64_t)d.sx * d.sy));
	// neighbour markers; a quad that would reach outside the grid belongs to border cells only, which do nothing
	const int ml = idx0 > 0 ? tmp[idx0 - 1] : 0, mr = idx0 + 4 < d.n ? tmp[idx0 + 4] : 0;
	I4u yp = {0, 0, 0, 0}, ym = {0, 0, 0, 0};
	int4 zp = make_int4(0, 0, 0, 0), zm = make_int4(0, 0, 0, 0);
	if (idx0 + Y + 3 < d.n) yp = *(const I4u*)(tmp + idx0 + Y);
	if (idx0 - Y >= 0) ym = *(const I4u*)(tmp + idx0 - Y);
	const bool zal = (Z & 3) == 0;
	if (idx0 + Z + 3 < d.n) {
		if (zal) zp = *(const int4*)(tmp + idx0 + Z);
		else {
			const I4u t = *(const I4u*)(tmp + idx0 + Z);
			zp = make_int4(t.x, t.y, t.z, t.w);
		}
	}
	if (idx0 - Z >= 0) {
		if (zal) zm = *(const int4*)(tmp + idx0 - Z);
		else {
			const I4u t = *(const I4u*)(tmp + idx0 - Z);
			zm = make_int4(t.x, t.y, t.z, t.w);
		}
	}
	const int nxp[4] = {m[1], m[2], m[3], mr}, nxm[4] = {ml, m[0], m[1], m[2]};
	const int nyp[4] = {yp.x, yp.y, yp.z, yp.w}, nym[4] = {ym.x, ym.y, ym.z, ym.w};
	const int nzp[4] = {zp.x, zp.y, zp.z, zp.w}, nzm[4] = {zm.x, zm.y, zm.z, zm.w};
#pragma unroll
	for (int c = 0; c < 4; c++) {
		const int64_t idx = idx0 + c;
		const bool interior = i >= 1 && i < d.sx - 1 && j >= 1 && j < d.sy - 1 && k >= 1 && k < d.sz - 1;
		if (interior && m[c] == 0) {
			int nbs = 0;
			float avg = 0.f;
			if (nxp[c] == dd) { avg += velc[idx + 1]; nbs++; }
			if (nxm[c] == dd) { avg += velc[idx - 1]; nbs++; }
			if (nyp[c] == dd) { avg += velc[idx + Y]; nbs++; }
			if (nym[c] == dd) { avg += velc[idx - Y]; nbs++; }
			if (nzp[c] == dd) { avg += velc[idx + Z]; nbs++; }
			if (nzm[c] == dd) { avg += velc[idx - Z]; nbs++; }
			if (nbs > 0) {
				tmp[idx] = dd + 1;
				velc[idx] = avg / (float)nbs;
			}
		}
		// next cell of the quad
		if (++i == d.sx) {
			i = 0;
			if (++j == d.sy) {
				j = 0;
				k++;
			}
		}
	}
}
// knExtrapolateIntoBnd, fastmarch.cpp:261-300 (bnd = 0: every cell; only border cells change)
__global__ void __launch_bounds__(BLOCK)
k_extrap_into_bnd(Dim d, const int32_t* __restrict__ flags, float* __restrict__ vel, const float* __restrict__ velTmp) {
	CELL_IJK(d)
	const int64_t n = d.n;
	int cnt = 0;
	float v0 = 0.f, v1 = 0.f, v2 = 0.f;
	const bool isObs = flags[idx] & MF_OBSTACLE;
#define TAKE(q, comp, cond)                                            \
	{                                                                  \
		v0 = velTmp[q];                                                \
		v1 = velTmp[n + (q)];                                          \
		v2 = velTmp[2 * n + (q)];                                      \
		if (isObs && (cond)) { if (comp == 0) v0 = 0.f; else if (comp == 1) v1 = 0.f; else v2 = 0.f; } \
		cnt++;                                                         \
	}
	if (i == 0) TAKE(idx + 1, 0, v0 < 0.f)
	else if (i == d.sx - 1) TAKE(idx - 1, 0, v0 > 0.f)
	if (j == 0) TAKE(idx + d.Y, 1, v1 < 0.f)
	else if (j == d.sy - 1) TAKE(idx - d.Y, 1, v1 > 0.f)
	if (d.is3d) {
		if (k == 0) TAKE(idx + d.Z, 2, v2 < 0.f)
		else if (k == d.sz - 1) TAKE(idx - d.Z, 2, v2 > 0.f)
	}
#undef TAKE
	if (cnt > 0) {
		const float fc = (float)cnt;
		vel[idx] = v0 / fc;
		vel[n + idx] = v1 / fc;
		vel[2 * n + idx] = v2 / fc;
	}
}
// extrapolateMACFromWeight, fastmarch.cpp:378-430
__global__ void __launch_bounds__(BLOCK) k_weight_reset(Dim d, float* __restrict__ wc) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	if (wc[idx] > 0.f) wc[idx] = 1.0f;
}
__global__ void __launch_bounds__(BLOCK)
k_extrap_weight(Dim d, float* __restrict__ velc, float* __restrict__ wc, int dd) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	if (wc[idx] != 0.f) return;
	const int64_t nb[6] = {1, -1, d.Y, -d.Y, d.Z, -d.Z};
	const int nn = d.is3d ? 6 : 4;
	int nbs = 0;
	float avg = 0.f;
	const float fd = (float)dd;
	for (int n = 0; n < nn; n++)
		if (wc[idx + nb[n]] == fd) {
			avg += velc[idx + nb[n]];
			nbs++;
		}
	if (nbs > 0) {
		wc[idx] = (float)(dd + 1);
		velc[idx] = avg / (float)nbs;
	}
}
// markFluidCells, flip.cpp:142-188
__global__ void __launch_bounds__(BLOCK) k_clear_fluid_flags(Dim d, int32_t* __restrict__ flags) {
	const int64_t idx = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (idx >= d.n) return;
	const int f = flags[idx];
	if (f & MF_FLUID) flags[idx] = (f | MF_EMPTY) & ~MF_FLUID;
}
__global__ void __launch_bounds__(BLOCK)
k_mark_fluid(Dim d, int32_t* __restrict__ flags, int64_t np, int64_t ps, const float* __restrict__ pos,
             const int32_t* __restrict__ pflag, const int32_t* __restrict__ ptype, int exclude) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	if ((pflag[p] & MF_PDELETE) || (ptype && (ptype[p] & exclude))) return;
	const int i = (int)pos[p], j = (int)pos[ps + p];   // toVec3i truncation
	const int k = (int)pos[2 * ps + p] - d.zoff;       // plane inside the slab window (global coordinates; identity without one)
	if (i < 0 || j < 0 || k < 0 || i >= d.sx || j >= d.sy || k >= d.sz) return;
	const int64_t idx = (int64_t)i + d.Y * j + d.Z * k;
	const int f = flags[idx];
	// every particle of a cell writes the same value, so concurrent marking is order independent
	if (f & MF_EMPTY) flags[idx] = (f | MF_FLUID) & ~MF_EMPTY;
}
// knSetNbObstacle, flip.cpp:149-164
__global__ void __launch_bounds__(BLOCK)
k_set_nb_obstacle(Dim d, int32_t* __restrict__ nflags, const int32_t* __restrict__ flags, const float* __restrict__ phiObs) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	if (phiObs[idx] > 0.f) return;
	if (!(flags[idx] & MF_EMPTY)) return;
	bool set = false;
	if ((flags[idx - 1] & MF_FLUID) && (phiObs[idx + 1] <= 0.f)) set = true;
	if ((flags[idx + 1] & MF_FLUID) && (phiObs[idx - 1] <= 0.f)) set = true;
	if ((flags[idx - d.Y] & MF_FLUID) && (phiObs[idx + d.Y] <= 0.f)) set = true;
	if ((flags[idx + d.Y] & MF_FLUID) && (phiObs[idx - d.Y] <= 0.f)) set = true;
	if (d.is3d) {
		if ((flags[idx - d.Z] & MF_FLUID) && (phiObs[idx + d.Z] <= 0.f)) set = true;
		if ((flags[idx + d.Z] & MF_FLUID) && (phiObs[idx - d.Z] <= 0.f)) set = true;
	}
	if (set) nflags[idx] = (flags[idx] | MF_FLUID) & ~MF_EMPTY;
}

extern "C" {
int mf_extrapolate_mac_simple(int sx, int sy, int sz, const int32_t* flags, float* vel, int distance, int intoObs,
                              int32_t* tmp, float* velTmp, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	const int dim = d.is3d ? 3 : 2;
	for (int c = 0; c < dim; c++) {
		hipLaunchKernelGGL(k_extrap_mark, dim3(nblk(d)), dim3(BLOCK), 0, st, d, flags, tmp, c, intoObs);
		const bool quads = d.is3d && ((((uintptr_t)tmp) & 15) == 0);
		for (int dd = 1; dd < 1 + distance; dd++) {
			if (quads)
				hipLaunchKernelGGL(k_extrap_simple4, dim3((unsigned)((d.n / 4 + BLOCK) / BLOCK)), dim3(BLOCK), 0, st, d, vel + c * d.n, tmp, dd);
			else
				hipLaunchKernelGGL(k_extrap_simple, dim3(nblk(d)), dim3(BLOCK), 0, st, d, vel + c * d.n, tmp, dd);
		}
	}
	MF_HIP(hipMemcpyAsync(velTmp, vel, sizeof(float) * 3 * d.n, hipMemcpyDeviceToDevice, st));
	hipLaunchKernelGGL(k_extrap_into_bnd, dim3(nblk(d)), dim3(BLOCK), 0, st, d, flags, vel, velTmp);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_extrapolate_mac_from_weight(int sx, int sy, int sz, float* vel, float* weight, int distance, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	const int dim = d.is3d ? 3 : 2;
	for (int c = 0; c < dim; c++) {
		hipLaunchKernelGGL(k_weight_reset, dim3(nblk(d)), dim3(BLOCK), 0, st, d, weight + c * d.n);
		for (int dd = 1; dd < 1 + distance; dd++)
			hipLaunchKernelGGL(k_extrap_weight, dim3(nblk(d)), dim3(BLOCK), 0, st, d, vel + c * d.n, weight + c * d.n, dd);
	}
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_mark_fluid_cells(int sx, int sy, int sz, int32_t* flags, int64_t np, int64_t ps, const float* pos, const int32_t* pflag,
                        const int32_t* ptype, int exclude, const float* phiObs, int32_t* ftmp, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	hipLaunchKernelGGL(k_clear_fluid_flags, dim3(nblk(d)), dim3(BLOCK), 0, st, d, flags);
	if (np > 0)
		hipLaunchKernelGGL(k_mark_fluid, dim3((unsigned)((np + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, d, flags, np, ps, pos, pflag, ptype, exclude);
	if (phiObs) {
		// FlagGrid tmp(flags); knSetNbObstacle(tmp, flags, phiObs); flags.swap(tmp)
		MF_HIP(hipMemcpyAsync(ftmp, flags, sizeof(int32_t) * d.n, hipMemcpyDeviceToDevice, st));
		hipLaunchKernelGGL(k_set_nb_obstacle, dim3(nblk(d)), dim3(BLOCK), 0, st, d, ftmp, flags, phiObs);
		MF_HIP(hipMemcpyAsync(flags, ftmp, sizeof(int32_t) * d.n, hipMemcpyDeviceToDevice, st));
	}
	MF_LAUNCH_CHECK();
	return 0;
}
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
