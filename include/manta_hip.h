/*
 * manta_hip.h -- C ABI of the MI355X (gfx950) hot-path library `libmanta_hip.so`.
 *
 * The reference (zoharl3/mantaflow, /root/reference) has NO C ABI for this path: its boundary is the
 * CPython module "manta" compiled into the executable (source/pwrapper/registry.cpp:21,488) and every
 * plugin is reached through a generated `_W_n(self,args,kwds)` wrapper (preprocessor/codegen_python.cpp:32-58).
 * What a maintainer would bind instead is therefore the set of KERNEL()/PYTHON() functions below; each
 * entry point cites the reference function it replaces.  INTEGRATION.md shows the binding stub.
 *
 * The same signatures are implemented three times:
 *   libmanta_hip.so     (mantaflow_amd/csrc)  the product: HIP kernels, every pointer is a DEVICE pointer
 *   libmanta_oracle.so  (oracle/)             plain-C restatement, HOST pointers, test infrastructure only
 *   libmanta_ref.so     (oracle/_ref)         the reference's own C++ behind a shim, HOST pointers, tests only
 *
 * Conventions
 *   - all functions return 0 on success, non-zero on error; mf_last_error() gives the message
 *     (reference: Manta::Error thrown by errMsg/assertMsg, general.h:42-77, surfaced as RuntimeError).
 *   - pointers are BORROWED; nothing is allocated that the caller must free, except through mf_ws_*.
 *   - `stream` is a hipStream_t (NULL = default stream); CPU libraries ignore it.  Calls are asynchronous
 *     on the stream unless they return a scalar through a host pointer (those synchronise the stream).
 *   - grids are dense, x fastest: idx = i + sx*(j + sy*k)  (reference grid.h:77).  2-D grids have sz == 1.
 *   - Real grids: float[N]; FlagGrid/IntGrid: int32[N]  (N = sx*sy*sz).
 *   - Vec3 / MAC grids are STRUCTURE-OF-ARRAYS: float[3][N] (x plane, y plane, z plane).  The reference
 *     stores AoS Vec3 (vectorbase.h:69); the AoS view only exists at the numpy bridge (python side).
 *   - particle positions / Vec3 pdata are SoA as well: float[3][cap]; component stride `pstride` (>= np)
 *     is passed explicitly.  particle flags: int32[np] (particle.h:34-43: PNEW=1, PDELETE=1<<10 ...).
 */
#ifndef MANTA_HIP_H
#define MANTA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FlagGrid::CellType, grid.h:306-320 */
enum { MF_FLUID = 1, MF_OBSTACLE = 2, MF_EMPTY = 4, MF_INFLOW = 8, MF_OUTFLOW = 16, MF_OPEN = 32, MF_STICK = 64 };
/* ParticleBase::SystemType status bits, particle.h:34-43 */
enum { MF_PNEW = 1, MF_PDELETE = 1 << 10 };
/* IntegrationMode, util/integrator.h:22 */
enum { MF_INT_EULER = 0, MF_INT_RK2 = 1, MF_INT_RK4 = 2 };
/* GridCgInterface::PreconditionType, conjugategrad.h:27 */
enum { MF_PC_NONE = 0, MF_PC_ICP = 1, MF_PC_MICP = 2, MF_PC_MGP = 3 };

const char* mf_last_error(void);
/* "hip" | "oracle" | "reference" -- which implementation answered (tests assert on this) */
const char* mf_backend(void);
/* ABI revision of this header.  It changes whenever an existing entry point changes its arguments (a binding built against another
 * revision would pass, e.g., the stream pointer where an int is now expected): a loader compares mf_abi_version() of the library
 * it opened with the MF_ABI_VERSION it was built against and refuses a mismatch (mantaflow_amd/_lib.py does).
 *   1  round 1
 *   2  round 2: mf_semi_lagrange_{real,vec3,mac}, mf_interpolate_grid, mf_interpolate_mac_grid take `int orderSpace` before `stream`;
 *      mf_apply_noise_vec3 takes four uv arguments; mf_set_mic_blocking / mf_set_mic_blocking_x replaced by mf_mic_init_blocked
 *   3  round 3: mf_abi_version itself; mf_set_mic_mode knows "rows" and "levels" only; mf_cg_slab_after_dp / _after_zr: the pressure
 *      update moved from the former to the latter (arguments changed); mf_pack_matrix takes A0; mf_solve_pressure_fused */
#define MF_ABI_VERSION 3
int mf_abi_version(void);

/* z-slab window (multi-GPU, no reference counterpart): subsequent calls on this thread treat every grid as the
 * planes [zoff, zoff+sz) of a global grid with gsz planes -- positions handed to the interpolators are global
 * coordinates, so a slab reproduces the undivided domain bit for bit.  (0, 0) restores the default (whole domain). */
int mf_set_slab_window(int zoff, int gsz);
/* the window of the SOURCE grid of the calls that read a grid of another size (mf_interpolate_grid, mf_interpolate_mac_grid,
 * the weight grid of mf_apply_noise_vec3): in a two-resolution scene each solver's slab has its own window */
int mf_set_slab_window_source(int zoff, int gsz);

/* How the MIC(0) sweeps are parallelised on the GPU (no reference counterpart; every mode gives the same bits as the serial sweeps of
 * conjugategrad.cpp:66-97, 135-159): "rows" (default for 3D: row-streaming dataflow sweeps, one 8 x 8 bundle of x-rows per workgroup,
 * one launch per sweep) or "levels" (one launch per hyperplane of 8^3 tiles: no waiting between workgroups); NULL or "" = back to the
 * default / MF_MIC_MODE.  The mode is taken by the next mf_mic_init* and stays with the system it registers.  Returns 0, or -1 for an
 * unknown name.  The oracle accepts and ignores it. */
int mf_set_mic_mode(const char* name);
/* Synchronises the stream and reports whether any MIC sweep since the last check gave up waiting for a neighbouring
 * workgroup (a deadlock guard of the single-launch sweeps; never seen in practice).  mf_cg_solve checks by itself; callers
 * that drive mf_mic_apply directly call this once per solve.  The oracle returns 0. */
int mf_mic_check(void* stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise grid ops used inside the CG loop and by scenes
 * ---------------------------------------------------------------------------------------------- */
/* Grid<T>::clear / setConst, grid.cpp:95-97,279 */
int mf_fill_f32(int64_t n, float* a, float value, void* stream);
int mf_fill_i32(int64_t n, int32_t* a, int32_t value, void* stream);
/* Grid<T>::copyFrom, grid.cpp:228-233 (n in elements of 4 bytes) */
int mf_copy_f32(int64_t n, float* dst, const float* src, void* stream);
/* gridScaledAdd<Real,Real>: me += factor*other, grid.h:514 */
int mf_grid_scaled_add(int64_t n, float* me, const float* other, float factor, void* stream);
/* UpdateSearchVec: dst = src + factor*dst, conjugategrad.cpp:193-196 */
int mf_update_search_vec(int64_t n, float* dst, const float* src, float factor, void* stream);
/* GridDotProduct: sum of fp32 products accumulated in fp64, conjugategrad.cpp:175-178 */
int mf_grid_dot(int64_t n, const float* a, const float* b, double* result_host, void* stream);
/* GridSumSqr, commonkernels.h:32-35 */
int mf_grid_sum_sqr(int64_t n, const float* a, double* result_host, void* stream);
/* Grid<Real>::getMaxAbs = max(|CompMinReal|,|CompMaxReal|), grid.cpp:185-196,356-360 */
int mf_grid_max_abs(int64_t n, const float* a, float* result_host, void* stream);
/* Grid<Real>::getMin / getMax */
int mf_grid_min_max(int64_t n, const float* a, float* min_host, float* max_host, void* stream);
/* Grid<Vec3>::getMaxAbs = sqrt(max normSquare), grid.cpp:198-224,367-369 (SoA input) */
int mf_grid_max_abs_vec3(int64_t n, const float* a, float* result_host, void* stream);
/* knGridStomp<Vec3>: v[c] < th ? 0 : v[c], grid.cpp:247-249 (applied to all 3n scalars) */
int mf_grid_stomp(int64_t n, float* a, float threshold, void* stream);
/* knGridSafeDiv: me = other ? me/other : me, grid.cpp:242, general.h:150 */
int mf_grid_safe_divide(int64_t n, float* me, const float* other, void* stream);
/* knGridAddConstReal / knGridMultConst / knGridClamp, grid.cpp:239-245 */
int mf_grid_add_const(int64_t n, float* me, float v, void* stream);
int mf_grid_mult_const(int64_t n, float* me, float v, void* stream);
int mf_grid_clamp(int64_t n, float* me, float lo, float hi, void* stream);
/* gridAdd / gridSub / gridMult, grid.h:508-512 */
int mf_grid_add(int64_t n, float* me, const float* other, void* stream);
int mf_grid_sub(int64_t n, float* me, const float* other, void* stream);
int mf_grid_mult(int64_t n, float* me, const float* other, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pressure projection
 * ---------------------------------------------------------------------------------------------- */
/* ApplyMatrix / ApplyMatrix2D (sz==1), conjugategrad.h:118-151.  28 B/cell algorithmic traffic. */
int mf_apply_matrix(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src,
                    const float* A0, const float* Ai, const float* Aj, const float* Ak, void* stream);
/* Optional accelerator for repeated mf_apply_matrix calls on one matrix: packs flags + Ai + Aj + Ak into one byte per cell
 * (valid only if every off-diagonal is exactly +0 or -1, checked here; one stream synchronisation).  Later mf_apply_matrix
 * calls with exactly these pointers read 13 instead of 28 B per cell; results are bit-identical.  With A0 (nullable; revision 3)
 * the diagonal travels in the same byte when every fluid cell's A0 is a small non-negative integer (MakeLaplaceMatrix's count of
 * non-obstacle neighbours; bit patterns compared) -- 9 B per cell for calls that pass this A0.  The grids must not change
 * while the packed bytes are in use; a new call replaces them. */
int mf_pack_matrix(int sx, int sy, int sz, const int32_t* flags, const float* A0, const float* Ai, const float* Aj, const float* Ak,
                   void* stream);
/* MakeLaplaceMatrix, conjugategrad.h:154-187.  A0..Ak must be zeroed by the caller (fresh temp grids).
 * fractions: nullable SoA MAC grid. */
int mf_make_laplace_matrix(int sx, int sy, int sz, const int32_t* flags, float* A0, float* Ai, float* Aj,
                           float* Ak, const float* fractions, void* stream);
/* matrix set-up of cgSolveDiffusion, conjugategrad.cpp:359-375, applied to the Laplace matrix of an all-fluid dummy flag grid
 * (mf_make_laplace_matrix on TypeFluid everywhere): obstacle cells get the identity row (Ai=Aj=Ak=0, A0=1), all other cells
 * A* *= alpha, A0 += 1.  The solve itself is mf_cg_solve without preconditioner (GridCg<ApplyMatrix/2D>). */
int mf_diffusion_matrix(int sx, int sy, int sz, const int32_t* flags, float* A0, float* Ai, float* Aj, float* Ak,
                        float alpha, void* stream);
/* MakeRhs, plugin/pressure.cpp:32-84.  rhs border cells are left untouched (bnd=1 kernel).
 * perCellCorr/fractions/obvel/phi/curv nullable.  cnt/sum returned through host pointers (nullable). */
int mf_make_rhs(int sx, int sy, int sz, const int32_t* flags, float* rhs, const float* vel,
                const float* perCellCorr, const float* fractions, const float* obvel,
                const float* phi, const float* curv, float surfTens, float gfClamp,
                int32_t* cnt_host, double* sum_host, void* stream);
/* ApplyGhostFluidDiagonal, plugin/pressure.cpp:136-151 */
int mf_apply_ghost_fluid_diagonal(int sx, int sy, int sz, float* A0, const int32_t* flags, const float* phi,
                                  float gfClamp, void* stream);
/* knCorrectVelocity, plugin/pressure.cpp:87-109 */
int mf_correct_velocity(int sx, int sy, int sz, const int32_t* flags, float* vel, const float* pressure,
                        void* stream);
/* knCorrectVelocityGhostFluid, plugin/pressure.cpp:154-187 (curv nullable) */
int mf_correct_velocity_ghost_fluid(int sx, int sy, int sz, float* vel, const int32_t* flags,
                                    const float* pressure, const float* phi, float gfClamp,
                                    const float* curv, float surfTens, void* stream);
/* knReplaceClampedGhostFluidVels, plugin/pressure.cpp:198-214 */
int mf_replace_clamped_ghost_fluid_vels(int sx, int sy, int sz, float* vel, const int32_t* flags,
                                        const float* pressure, const float* phi, float gfClamp, void* stream);
/* CountEmptyCells, plugin/pressure.cpp:217-220 */
int mf_count_empty_cells(int64_t n, const int32_t* flags, int32_t* result_host, void* stream);
/* fixPressure, plugin/pressure.cpp:226-246 */
int mf_fix_pressure(int sx, int sy, int sz, int64_t fixPidx, float value, float* rhs, float* A0, float* Ai,
                    float* Aj, float* Ak, void* stream);

/* CONCURRENCY CONTRACT of the MIC sweeps (mf_mic_init*, mf_mic_apply*, mf_cg_solve, mf_pack_matrix): per device the library keeps ONE
 * set of sweep state (ticket counters, generation tags, face exchange buffers, the bundle map and packed bytes of the last
 * mf_mic_init).  Calls that use it must be issued from one host thread at a time and on one stream per device; two solves in flight
 * on the same device (two streams, or two host threads) would share tickets and tags.  Different devices are independent.
 *
 * InitPreconditionModifiedIncompCholesky2, conjugategrad.cpp:66-97 (3-D only).  Also records, for exactly these grids, which
 * 8x8 bundles of x-rows hold no fluid cell and have no coupling into them: mf_mic_apply leaves those out when it is called
 * with the same flags / Aprecond / Aj / Ak pointers (as GridCg does); with other pointers it sweeps everything.  Changing the
 * contents of these grids between mf_mic_init and mf_mic_apply is a caller error (the reference's Aprecond would be stale too). */
int mf_mic_init(int sx, int sy, int sz, const int32_t* flags, float* Aprecond, const float* A0,
                const float* Ai, const float* Aj, const float* Ak, void* stream);
/* Multi-GPU only (no reference counterpart): mf_mic_init for a preconditioner the caller has cut into independent blocks of
 * `rows_j` grid rows along y and `cells_x` cells along x (non-negative multiples of 8; 0 = uncut) by zeroing the Aj / Ai
 * coupling across the block faces in the copies passed here -- the block-Jacobi form the z-slab solver already uses across
 * slabs.  The blocking is remembered WITH this system (the flags / Aprecond / Aj / Ak pointers): mf_mic_apply called with the
 * same pointers skips the hand-offs across the block faces (the values exchanged there are multiplied by 0 anyway), which
 * shortens the dependency chain; any other system is swept uncut.  The arithmetic is unchanged: results equal the serial
 * sweep over the same (cut) coefficients bit for bit.  The oracle ignores the two numbers. */
int mf_mic_init_blocked(int sx, int sy, int sz, const int32_t* flags, float* Aprecond, const float* A0,
                        const float* Ai, const float* Aj, const float* Ak, int rows_j, int cells_x, void* stream);
/* ApplyPreconditionModifiedIncompCholesky2, conjugategrad.cpp:135-159 (3-D only).
 * dst keeps its previous content in non-fluid cells, exactly like the reference. */
int mf_mic_apply(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* var1,
                 const float* Aprecond, const float* Ai, const float* Aj, const float* Ak, void* stream);
/* mf_mic_apply followed by dot_dev[0] = GridDotProduct(dst, var1) over the whole grid (fp32 products, fp64 sum), the sum
 * coming out of the backward sweep when the active sweep mode can fuse it */
int mf_mic_apply_dot_dev(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* var1,
                         const float* Aprecond, const float* Ai, const float* Aj, const float* Ak, double* dot_dev,
                         void* stream);

/* GridCg<ApplyMatrix>::doInit + iterate loop, conjugategrad.cpp:210-307, driven like
 * solvePressureSystem's loop (plugin/pressure.cpp:438-441).
 *   dst(pressure), residual, search, tmp : work grids, overwritten (dst = solution; the contents of residual / search / tmp /
 *   Aprecond after the call are unspecified: with the MIC preconditioner and a row length that is not a multiple of 8 the loop
 *   runs on an internal copy of the system whose rows are padded with obstacle cells -- same iterates, 16-byte rows)
 *   pc = MF_PC_NONE | MF_PC_MICP; Aprecond: grid for the MIC factor (pca0), unused for PC_NONE
 *   Liquid scenes (MIC-PCG): where 8 x 8 bundles of x-rows hold no fluid cell, or the fluid keeps to a part of the x-range, and rhs
 *   and the incoming work grids are +0 there -- checked on the device, once per call -- the kernels of an iteration leave those
 *   cells out: they stay +0, as they do in the reference.  A rhs that is not zero there is solved without the shortcut.
 *   out_host[0] = iterations done, out_host[1] = final residual norm (mResNorm), out_host[2] = mSigma
 * returns non-zero with message "The CG solver diverged" when resNorm !< 1e35 (conjugategrad.cpp:288-295). */
int mf_cg_solve(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* rhs, float* residual,
                float* search, float* tmp, const float* A0, const float* Ai, const float* Aj, const float* Ak,
                float* Aprecond, int pc, float accuracy, int maxIter, int useL2Norm,
                float* out_host, void* stream);
/* What the last mf_cg_solve / mf_solve_pressure_fused of this thread did with a liquid scene (see above): out[0] = 1 when its streaming
 * kernels were allowed to leave the fluid-free bundles out, out[1] / out[2] = first cell and number of cells of the x-range its MIC
 * sweeps kept to (0 / 0: whole rows).  For tests and diagnostics; the oracle reports zeros (it has no shortcut). */
int mf_cg_last_shortcut(int32_t* out);

/* ------------------------------------------------------------------------------------------------
 * Advection (plugin/advection.cpp)
 * ---------------------------------------------------------------------------------------------- */
/* SemiLagrange<Real>, advection.cpp:25-42 (orderTrace 1|2: first-order / midpoint back trace; orderSpace 1|2: linear interpolation,
 * util/interpol.h, or cubic, util/interpolHigh.h -- Grid::getInterpolatedHi, grid.h:153-159).  KERNEL(bnd=1) into a fresh temp grid: the
 * border cells of dst are written as 0 (mf_semi_lagrange_vec3 / _mac alike), the caller need not clear dst */
int mf_semi_lagrange_real(int sx, int sy, int sz, const float* vel, float* dst, const float* src, float dt,
                          int orderTrace, int orderSpace, void* stream);
/* SemiLagrange<Vec3> (cell-centred Vec3 grid, SoA) */
int mf_semi_lagrange_vec3(int sx, int sy, int sz, const float* vel, float* dst, const float* src, float dt,
                          int orderTrace, int orderSpace, void* stream);
/* SemiLagrangeMAC, advection.cpp:45-78 (orderSpace 2: MACGrid::getInterpolatedComponentHi -> interpolCubicMAC) */
int mf_semi_lagrange_mac(int sx, int sy, int sz, const float* vel, float* dst, const float* src, float dt,
                         int orderTrace, int orderSpace, void* stream);
/* MacCormackCorrect<Real|Vec3> KERNEL(idx), advection.cpp:82-92; ncomp = 1 (Real) or 3 (Vec3 SoA) */
int mf_maccormack_correct(int sx, int sy, int sz, int ncomp, const int32_t* flags, float* dst, const float* old,
                          const float* fwd, const float* bwd, float strength, void* stream);
/* MacCormackCorrectMAC<Vec3>(isMAC=true), advection.cpp:95-116 */
int mf_maccormack_correct_mac(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* old,
                              const float* fwd, const float* bwd, float strength, void* stream);
/* MacCormackClamp<Real|Vec3>, advection.cpp:242-268 + doClampComponent :145-187 */
int mf_maccormack_clamp(int sx, int sy, int sz, int ncomp, const int32_t* flags, const float* vel, float* dst,
                        const float* orig, const float* fwd, float dt, int clampMode, void* stream);
/* MacCormackClampMAC, advection.cpp:271-288 + doClampComponentMAC :192-236 */
int mf_maccormack_clamp_mac(int sx, int sy, int sz, const int32_t* flags, const float* vel, float* dst,
                            const float* orig, const float* fwd, float dt, int clampMode, void* stream);
/* MacCormackCorrect + MacCormackClamp in ONE pass (advection.cpp:82-92 + 242-268, MAC: 95-116 + 271-288): the clamp only looks at
 * the corrected value of its own cell, so the intermediate grid of the two-kernel sequence never has to exist.  Same result, bit
 * for bit, as mf_maccormack_correct followed by mf_maccormack_clamp (dst receives the corrected value in the border cells the
 * clamp does not visit).  dst must not alias orig / fwd / bwd. */
int mf_maccormack_correct_clamp(int sx, int sy, int sz, int ncomp, const int32_t* flags, const float* vel, float* dst,
                                const float* orig, const float* fwd, const float* bwd, float strength, float dt, int clampMode,
                                void* stream);
int mf_maccormack_correct_clamp_mac(int sx, int sy, int sz, const int32_t* flags, const float* vel, float* dst, const float* orig,
                                    const float* fwd, const float* bwd, float strength, float dt, int clampMode, void* stream);
/* applyOutflowBC = extrapolateVelConvectiveBC + copyChangedVels, advection.cpp:347-392.
 * velDst is a zeroed MAC scratch grid supplied by the caller. timeStep is the solver dt. */
int mf_apply_outflow_bc(int sx, int sy, int sz, const int32_t* flags, float* vel, const float* velPrev,
                        float* velDst, float timeStep, void* stream);

/* ------------------------------------------------------------------------------------------------
 * FLIP particle <-> grid transfers (plugin/flip.cpp, particle.h, util/interpol.h, util/integrator.h)
 * ptype nullable; a particle is skipped when (ptype[idx] & exclude).
 * ---------------------------------------------------------------------------------------------- */
/* mapPartsToMAC, flip.cpp:637-661: clears vel+weight, scatters (knMapLinearVec3ToMACGrid :619-633 ->
 * setInterpolMAC interpol.h:166-213), weight.stomp(1e-6), vel.safeDivide(weight), velOld.copyFrom(vel).
 * deterministic != 0 : contributions are summed in particle-index order per cell (bit-identical to the
 * reference's single-threaded scatter); 0 : fp32 atomics (order-dependent last bits). */
int mf_map_parts_to_mac(int sx, int sy, int sz, float* vel, float* velOld, float* weight,
                        int64_t np, int64_t pstride, const float* pos, const int32_t* pflag, const float* pvel,
                        const int32_t* ptype, int exclude, int deterministic, void* stream);
/* The two halves of mf_map_parts_to_mac, for a z-slab decomposition (SURVEY 8e "P2G scatter with reverse halo"): accum
 * clears vel+weight and scatters this rank's particles (positions in global coordinates, mf_set_slab_window) into the local
 * slab incl. one ghost plane per side; the caller adds the ghost-plane sums into the neighbour's boundary planes; finish
 * applies weight.stomp(1e-6), vel.safeDivide(weight), velOld.copyFrom(vel) to n3 = 3*cells scalars (velOld nullable). */
int mf_map_parts_to_mac_accum(int sx, int sy, int sz, float* vel, float* weight, int64_t np, int64_t pstride,
                              const float* pos, const int32_t* pflag, const float* pvel, const int32_t* ptype,
                              int exclude, int deterministic, void* stream);
int mf_map_parts_to_mac_finish(int64_t n3, float* vel, float* velOld, float* weight, void* stream);
/* mapMACToParts -> knMapLinearMACGridToVec3_PIC, flip.cpp:709-721 */
int mf_map_mac_to_parts(int sx, int sy, int sz, const float* vel, int64_t np, int64_t pstride, const float* pos,
                        const int32_t* pflag, float* pvel, const int32_t* ptype, int exclude, void* stream);
/* flipVelocityUpdate -> knMapLinearMACGridToVec3_FLIP, flip.cpp:724-742 */
int mf_flip_velocity_update(int sx, int sy, int sz, const float* vel, const float* velOld, int64_t np,
                            int64_t pstride, const float* pos, const int32_t* pflag, float* pvel,
                            float flipRatio, const int32_t* ptype, int exclude, void* stream);
/* APIC transfers, plugin/apic.cpp.  cpx/cpy/cpz: the affine rows, Vec3 pdata in SoA [3][pstride].
 * apicMapPartsToMAC :92-110 -> knApicMapLinearVec3ToMACGrid :19-90 (KERNEL(pts, single)): clears vel+mass, per particle
 * and face node  mass += w ; vel += w*v_c ; vel += w*dot(cp_c, node - pos), then mass.stomp(1e-6), vel.safeDivide(mass).
 * Summed in particle-index order per node (bit-identical to the reference's serial scatter).  mass is required here
 * (the Python layer passes a temporary MAC grid when the scene gives none). */
int mf_apic_map_parts_to_mac(int sx, int sy, int sz, float* vel, float* mass, int64_t np, int64_t pstride,
                             const float* pos, const int32_t* pflag, const float* pvel, const float* cpx,
                             const float* cpy, const float* cpz, const int32_t* ptype, int exclude, void* stream);
/* apicMapMACGridToParts :175-181 -> knApicMapLinearMACGridToVec3 :112-173: pvel = trilinear face sample, cp_c = its
 * gradient weights (gw = {-1, 1} on one axis) */
int mf_apic_map_mac_to_parts(int sx, int sy, int sz, const float* vel, int64_t np, int64_t pstride, const float* pos,
                             const int32_t* pflag, float* pvel, float* cpx, float* cpy, float* cpz,
                             const int32_t* ptype, int exclude, void* stream);
/* mapPartsToGrid / mapPartsToGridVec3, flip.cpp:663-687 (+ setInterpol interpol.h:96-113, knSafeDivReal
 * :607-615).  ncomp 1|3; target SoA; wtmp = zeroed Real scratch grid. */
int mf_map_parts_to_grid(int sx, int sy, int sz, int ncomp, float* target, float* wtmp, int64_t np,
                         int64_t pstride, const float* pos, const int32_t* pflag, const float* psrc,
                         int deterministic, void* stream);
/* mapGridToParts / mapGridToPartsVec3, flip.cpp:693-704 */
int mf_map_grid_to_parts(int sx, int sy, int sz, int ncomp, const float* source, int64_t np, int64_t pstride,
                         const float* pos, const int32_t* pflag, float* ptarget, void* stream);
/* ParticleSystem::advectInGrid, particle.h:526-550 (GridAdvectKernel :458-481, integratePointSet
 * integrator.h:26-78 incl. the fork's RK4 accumulation line 55, KnClampPositions :507-523,
 * KnDeleteInObstacle :485-491).  scratch: float[3*3*pstride] (x0, u, uTotal). pos/pflag updated in place. */
int mf_advect_in_grid(int sx, int sy, int sz, const int32_t* flags, const float* vel, int64_t np, int64_t pstride,
                      float* pos, int32_t* pflag, float dt, int integrationMode, int deleteInObstacle,
                      int stopInObstacle, int skipNew, const int32_t* ptype, int exclude, float* scratch,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * "next" rows: glue between advect and solve so a smoke step stays on the device
 * ---------------------------------------------------------------------------------------------- */
/* KnSetWallBcs (setWallBcs without fractions/phiObs), plugin/extforces.cpp:187-237,327-335; obvel nullable */
int mf_set_wall_bcs(int sx, int sy, int sz, const int32_t* flags, float* vel, const float* obvel, void* stream);
/* KnAddBuoyancy, plugin/extforces.cpp:73-88: strength = -gravity*dt/dx*coefficient computed by the caller */
int mf_add_buoyancy(int sx, int sy, int sz, const int32_t* flags, const float* density, float* vel,
                    float fx, float fy, float fz, void* stream);
/* KnApplyForce (addGravity), plugin/extforces.cpp:46-66; exclude: nullable Real grid (skip where < 0) */
int mf_apply_force(int sx, int sy, int sz, const int32_t* flags, float* vel, float fx, float fy, float fz,
                   const float* exclude, int additive, void* stream);

/* ------------------------------------------------------------------------------------------------
 * "next" rows: FLIP glue between particle->grid and grid->particle
 * ---------------------------------------------------------------------------------------------- */
/* extrapolateMACSimple without phiObs, fastmarch.cpp:231-376.  tmp: Int scratch grid, velTmp: MAC scratch grid */
int mf_extrapolate_mac_simple(int sx, int sy, int sz, const int32_t* flags, float* vel, int distance, int intoObs,
                              int32_t* tmp, float* velTmp, void* stream);
/* extrapolateMACFromWeight, fastmarch.cpp:378-430 (the weight grid is destroyed, like in the reference) */
int mf_extrapolate_mac_from_weight(int sx, int sy, int sz, float* vel, float* weight, int distance, void* stream);
/* markFluidCells, plugin/flip.cpp:142-188.  phiObs nullable; ftmp: Int scratch grid (used with phiObs only) */
int mf_mark_fluid_cells(int sx, int sy, int sz, int32_t* flags, int64_t np, int64_t pstride, const float* pos,
                        const int32_t* pflag, const int32_t* ptype, int exclude, const float* phiObs, int32_t* ftmp,
                        void* stream);

/* ParticleSystem::projectOutOfBnd -> KnProjectOutOfBnd, particle.h:579-604.  axis: bit0 'x', bit1 'X', bit2 'y',
 * bit3 'Y', bit4 'z', bit5 'Z' (the letters of the reference's `plane` string) */
int mf_project_out_of_bnd(int sx, int sy, int sz, int64_t np, int64_t pstride, float* pos, const int32_t* pflag,
                          float bnd, int axis, const int32_t* ptype, int exclude, void* stream);
/* pushOutofObs -> knPushOutofObs, plugin/flip.cpp:584-602 (getInterpolated grid.h:134, getGradient grid.h:556-573,
 * normalize vectorbase.h:421-434) */
int mf_push_out_of_obs(int sx, int sy, int sz, int64_t np, int64_t pstride, float* pos, const int32_t* pflag,
                       const float* phiObs, float shift, float thresh, const int32_t* ptype, int exclude, void* stream);

/* ------------------------------------------------------------------------------------------------
 * "next" rows: free-surface pieces of scenes/benchmark_dam.py
 * ---------------------------------------------------------------------------------------------- */
/* gridParticleIndex, plugin/flip.cpp:273-320: index(cell) = first slot of the cell in indexSys, indexSys = particle
 * indices ordered by (cell, particle index) -- the order the reference's serial counting sort produces.  Particles that
 * are inactive or outside the grid are skipped.  counter: Int scratch grid; keys/vals: int32[2*np] scratch each;
 * *n_indexed_host (nullable) receives the size the reference gives indexSys. */
int mf_grid_particle_index(int sx, int sy, int sz, int64_t np, int64_t pstride, const float* pos, const int32_t* pflag,
                           int32_t* indexSys, int32_t* index, int32_t* counter, int32_t* keys, int32_t* vals,
                           int64_t* n_indexed_host, void* stream);
/* unionParticleLevelset -> ComputeUnionLevelsetPindex + phi.setBound(0.5, 0), plugin/flip.cpp:322-363;
 * n_indexed = indexSys.size() */
int mf_union_particle_levelset(int sx, int sy, int sz, int64_t np, int64_t pstride, const float* pos,
                               const int32_t* indexSys, int64_t n_indexed, const int32_t* index, float* phi,
                               float radiusFactor, const int32_t* ptype, int exclude, void* stream);
/* extrapolateLsSimple, fastmarch.cpp:432-522 (knExtrapolateLsSimple, knSetRemaining).  tmp: Int scratch grid */
int mf_extrapolate_ls_simple(int sx, int sy, int sz, float* phi, int distance, int inside, int include_walls,
                             int32_t* tmp, void* stream);
/* setPartType -> KnSetPartType, plugin/ptsplugins.cpp:56-65 */
int mf_set_part_type(int sx, int sy, int sz, const int32_t* flags, int64_t np, int64_t pstride, const float* pos,
                     int32_t* ptype, int mark, int stype, int cflag, void* stream);
/* markIsolatedFluidCell -> knMarkIsolatedFluidCell, grid.cpp:987-1011 */
int mf_mark_isolated_fluid_cell(int sx, int sy, int sz, int32_t* flags, int mark, void* stream);
/* addForcePvel -> KnAddForcePvel (da = a*dt formed in fp32), plugin/ptsplugins.cpp:20-29 */
int mf_add_force_pvel(int64_t np, int64_t pstride, float* pvel, float ax, float ay, float az, float dt,
                      const int32_t* ptype, int exclude, void* stream);
/* updateVelocityFromDeltaPos -> KnUpdateVelocityFromDeltaPos (over_dt = 1.0/dt in double, rounded), ptsplugins.cpp:31-41 */
int mf_update_velocity_from_delta_pos(int64_t np, int64_t pstride, const float* pos, float* pvel, const float* xprev,
                                      float dt, const int32_t* ptype, int exclude, void* stream);
/* eulerStep -> KnStepEuler, ptsplugins.cpp:43-53 */
int mf_euler_step(int64_t np, int64_t pstride, float* pos, const float* pvel, float dt, const int32_t* ptype,
                  int exclude, void* stream);
/* Shape::computeLevelset -> generateLevelset (shapes.cpp): signed distance at the cell centres.
 *   kind 0 Box      (BoxSDF :178-229)       params: p0.xyz, p1.xyz
 *   kind 1 Sphere   (SphereSDF :303-307)    params: center.xyz, radius, scale.xyz
 *   kind 2 Cylinder (CylinderSDF :367-385)  params: center.xyz, radius, zaxis.xyz (already normalised), zlen
 * params_host: 12 floats read on the host at call time. */
int mf_shape_levelset(int sx, int sy, int sz, int kind, const float* params_host, float* phi, void* stream);
/* Shape::applyToGrid -> ApplyShapeToGrid<T> / ApplyShapeToMACGrid (shapes.cpp:40-69): cells (or, for a MAC grid, the three face
 * positions of a cell) inside the shape get `value`; cells flagged obstacle in respectFlags (nullable) are left alone.
 * gridkind 0 Real, 1 Vec3 (SoA), 2 MAC (SoA), 3 int (value_host[0] converted).  Shape parameters as for mf_shape_levelset. */
int mf_shape_apply_to_grid(int sx, int sy, int sz, int kind, const float* params_host, int gridkind, void* grid,
                           const float* value_host, const int32_t* respectFlags, void* stream);
/* resetOutflow, extforces.cpp:134-161: outflow cells get (flags | Empty) & ~Fluid, phi = 0.5, real = 0 (phi / real
 * nullable); active particles that lie inside the grid in an outflow cell are flagged PDELETE (np 0 / pos NULL: none).
 * The reference then compacts the particle array (doCompress); here deleted particles stay flagged and are skipped. */
int mf_reset_outflow(int sx, int sy, int sz, int32_t* flags, float* phi, float* real, int64_t np, int64_t pstride,
                     const float* pos, int32_t* pflag, void* stream);
/* LevelsetGrid::join -> KnJoin (min) / subtract -> KnSubtract (phi = -other where other < 0; flags nullable: only in
 * cells with flags & subtractType), levelset.cpp:107-118; Grid::setBound -> knSetBoundary grid.cpp:629-637 */
int mf_levelset_join(int64_t n, float* phi, const float* other, void* stream);
int mf_levelset_subtract(int64_t n, float* phi, const float* other, const int32_t* flags, int subtractType, void* stream);
int mf_grid_set_bound(int sx, int sy, int sz, float* grid, float value, int boundaryWidth, void* stream);

/* ------------------------------------------------------------------------------------------------
 * "next" rows: resampling between grids of different size (wavelet-turbulence up-res helpers)
 * ---------------------------------------------------------------------------------------------- */
/* interpolateGrid / interpolateGridVec3 -> knInterpolateGridTempl, grid.h:576-581, plugin/waveletturbulence.cpp:37-56
 * target(i,j,k) = source.getInterpolatedHi(Vec3(i,j,k) * sourceFactor + offset, orderSpace) (1 linear, 2 cubic); ncomp 1|3 (SoA planes);
 * sourceFactor / offset are the values calcGridSizeFactorMod (:27-34) produces. */
int mf_interpolate_grid(int tsx, int tsy, int tsz, float* target, int ssx, int ssy, int ssz, const float* source, int ncomp,
                        float sfx, float sfy, float sfz, float ox, float oy, float oz, int orderSpace, void* stream);
/* interpolateMACGrid -> KnInterpolateMACGrid, plugin/waveletturbulence.cpp:59-78: component c sampled at pos - 0.5 e_c */
int mf_interpolate_mac_grid(int tsx, int tsy, int tsz, float* target, int ssx, int ssy, int ssz, const float* source,
                            float sfx, float sfy, float sfz, float ox, float oy, float oz, int orderSpace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Wavelet noise (noisefield.{h,cpp}) and the noise-modulated smoke source of scenes/simpleplume.py
 * ---------------------------------------------------------------------------------------------- */
/* WaveletNoiseField::generateTile, noisefield.cpp:95-186: the 3 x 128^3 tile for `seed` (the reference's static
 * randomSeed is 13322223), generated on the host exactly as the reference does (MT19937 + Box-Muller in double, fp32
 * down/up-sampling filters) and stored to `tile` (device memory for the GPU library). */
int mf_noise_generate_tile(float* tile, int seed, void* stream);
/* mSeedOffset = RandomStream(fixedSeed).getVec3Norm(), noisefield.cpp:64-70 (fixedSeed -1 -> randomSeed + 123); host output */
int mf_noise_seed_offset(int fixedSeed, float* out3_host);
/* densityInflow -> KnApplyNoiseInfl, plugin/initplugins.cpp:27-43, with WaveletNoiseField::evaluate / WNoise,
 * noisefield.h:118-137, 313-336.  sdf = shape.computeLevelset().  params (host, 20 floats): [0..2] mGsInv, [3..5]
 * mSeedOffset, [6] getTime(), [7..9] posScale, [10..12] posOffset, [13] valOffset, [14] valScale, [15] clamp (0|1),
 * [16] clampNeg, [17] clampPos. */
int mf_density_inflow(int sx, int sy, int sz, const int32_t* flags, float* density, const float* sdf, const float* tile,
                      const float* params_host, float scale, float sigma, void* stream);

/* ------------------------------------------------------------------------------------------------
 * wavelet turbulence pieces of scenes/waveletTurbulence.py (plugin/waveletturbulence.cpp, plugin/extforces.cpp)
 * ---------------------------------------------------------------------------------------------- */
/* computeEnergy -> KnApplyComputeEnergy, waveletturbulence.cpp:180-194 */
int mf_compute_energy(int sx, int sy, int sz, const int32_t* flags, const float* vel, float* energy, void* stream);
/* vorticityConfinement, extforces.cpp:409-428: GetCentered (commonkernels.h:126-131), CurlOp (:38-47), GridNorm (:116-118),
 * KnConfForce, KnApplyForceField (extforces.cpp:24-43, additive, not MAC).  velCenter/curl/force: zeroed Vec3 scratch grids,
 * norm: Real scratch grid; strengthCell nullable. */
int mf_vorticity_confinement(int sx, int sy, int sz, float* vel, const int32_t* flags, float strength,
                             const float* strengthCell, float* velCenter, float* curl, float* norm, float* force, void* stream);
/* computeWaveletCoeffs -> WaveletNoiseField::computeCoefficients, noisefield.cpp:191-297; temp1/temp2: Real scratch grids */
int mf_compute_wavelet_coeffs(int sx, int sy, int sz, float* input, float* temp1, float* temp2, void* stream);
/* applyNoiseVec3 -> knApplyNoiseVec3, waveletturbulence.cpp:120-178: target += evaluateCurl(pos * scaleSpatial) * scale * w with
 * pos = (i,j,k)+0.5 or, with a uv grid (Grid<Vec3>, SoA), uv(i,j,k); w = weight(i,j,k).  When the uv grid (or, without one, the
 * weight grid) has another size than the target, both are read with getInterpolated at (i,j,k) * sourceFactor and the uv value is
 * divided by sourceFactor.  uv and weight must have the same size (usx.. / wsx.. both describe it).  params as for
 * mf_density_inflow; weight and uv nullable. */
int mf_apply_noise_vec3(int sx, int sy, int sz, const int32_t* flags, float* target, const float* tile,
                        const float* params_host, float scale, float scaleSpatial, const float* weight, int wsx, int wsy,
                        int wsz, const float* uv, int usx, int usy, int usz, void* stream);

/* ------------------------------------------------------------------------------------------------
 * device-scalar variants for the multi-GPU PCG (no reference counterpart: same arithmetic as mf_grid_dot /
 * mf_grid_max_abs / mf_grid_scaled_add / mf_update_search_vec, but the scalar results and factors live in device
 * memory, so a rank never waits for the host between a reduction, its all-gather and the update that uses it)
 * ---------------------------------------------------------------------------------------------- */
int mf_grid_dot_dev(int64_t n, const float* a, const float* b, double* out_dev, void* stream);
int mf_grid_max_abs_dev(int64_t n, const float* a, float* out_dev, void* stream);
int mf_grid_max_abs_dev_f64(int64_t n, const float* a, double* out_dev, void* stream);
/* scalar steps of GridCg::iterate (conjugategrad.cpp:250-291) on the all-gathered per-rank pairs
 * gathered[world][2] = {max|residual|, dot}: alpha = sigma / (Real)sum(dot) (0 if the sum is 0), also stored negated at
 * alpha_dev[1];
 * beta = (Real)sum(dot) / sigma, sigma := (Real)sum(dot), res = max over ranks.  Rows are combined in rank order.
 * state_dev (nullable) = int32[2] {stop, iteration}: the stopping test of GridCg::iterate (:262-272) evaluated on the
 * device so that the host need not read a scalar every iteration.  beta sets {1, iter} the first time res < accuracy and
 * {2, iter} when res is not < 1e35 (divergence); once stop != 0, beta leaves sigma/beta/res untouched and alpha returns 0,
 * which turns the vector updates of any further (speculatively queued) iteration into no-ops: pressure and residual keep
 * the values the reference stops with. */
int mf_cg_slab_alpha(const double* gathered, int world, const float* sigma_dev, float* alpha_dev,
                     const int32_t* state_dev, void* stream);
int mf_cg_slab_beta(const double* gathered, int world, float* sigma_dev, float* beta_dev, float* res_dev,
                    float accuracy, int iter, int32_t* state_dev, void* stream);
/* solvePressureSystem for the plain case -- a MakeLaplaceMatrix system (no fractions, no ghost-fluid diagonal, no pressure fixing) with
 * the plain MakeRhs (no perCellCorr / obvel / surface tension / enforceCompatibility), MIC preconditioner, 3D, sx % 8 == 0 -- without the
 * coefficient grids: ONE pass over flags and vel writes rhs (MakeRhs, pressure.cpp:32-84; every cell, 0 outside the fluid) and the packed
 * byte of every cell ({fluid, Ai / Aj / Ak == -1, integer A0}: MakeLaplaceMatrix, conjugategrad.h:154-187), the MIC factor is built from
 * those bytes (conjugategrad.cpp:66-97) and the PCG of mf_cg_solve runs on them.  Same arithmetic per cell, same iterates as
 * mf_make_rhs + mf_make_laplace_matrix + mf_cg_solve; rhs / residual / search / tmp / Aprecond are work grids that need not be cleared.
 * out_host as in mf_cg_solve.  HIP: fails for other grids (callers fall back to the three calls); the oracle composes the three calls. */
int mf_solve_pressure_fused(int sx, int sy, int sz, const int32_t* flags, const float* vel, float* pressure, float* rhs,
                            float* residual, float* search, float* tmp, float* Aprecond, float accuracy, int maxIter,
                            int useL2Norm, float* out_host, void* stream);
/* fused pieces of one slab PCG iteration.  `scalars` points to a device block laid out as
 *   float sigma, alpha, nalpha, beta, resNorm, dp, sigmaNew, accuracy; int32 iterations, done, diverged, useL2
 * (zero-initialised by the caller; mf_cg_slab_alpha writes alpha and nalpha = -alpha, mf_cg_slab_beta sigma / beta / resNorm):
 *   mf_apply_matrix_dot_dev : dst = A src (mf_apply_matrix) and dot_dev[0] = sum over the planes [k0, k1) of dst*src (fp32
 *                             products summed in fp64, GridDotProduct) -- a slab's own planes, not its ghosts
 *   mf_cg_slab_axpy2        : x += alpha*search ; residual += nalpha*tmp ; maxabs_dev[0] = max |residual| over the n cells */
int mf_apply_matrix_dot_dev(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src,
                            const float* A0, const float* Ai, const float* Aj, const float* Ak, int k0, int k1,
                            const void* scalars, double* dot_dev, void* stream);
int mf_cg_slab_axpy2(int64_t n, const void* scalars, float* x, const float* search, float* residual,
                     const float* tmp, double* maxabs_dev, void* stream);
/* me += (sign * factor_dev[0]) * other, sign = +-1 */
int mf_grid_scaled_add_dev(int64_t n, float* me, const float* other, const float* factor_dev, float sign, void* stream);
/* dst = src + factor_dev[0] * dst */
int mf_update_search_vec_dev(int64_t n, float* dst, const float* src, const float* factor_dev, void* stream);
/* The z-slab PCG iteration has three stretches of device work between its communication points (halo exchange of `search`, gather of
 * dot(A search, search), gather of {max|r|, dot(z, r)}).  These two entry points queue the second and the third stretch with ONE
 * call each; the host loop of a rank is per-iteration latency at N > 1.  As in mf_cg_solve, `x += alpha * search` rides on the search
 * update (`search` is read once for both; revision 3 of the ABI -- before, mf_cg_slab_after_dp updated x):
 *   mf_cg_slab_after_dp  : alpha from the gathered rows (0 once the stop state is set) ; residual += nalpha * tmp over the owned cells
 *                          with maxabs_dev[0] = max |residual| ; mf_mic_apply_dot_dev (tmp = M^-1 residual, dot_dev[0] = dot(tmp, residual))
 *   mf_cg_slab_after_zr  : beta and the stopping test from the gathered rows ; x += alpha * search over the owned cells (this iteration's
 *                          update, also when the iteration is the one that converged) ; unless stopped, search = tmp + beta * search
 * `scalars`: the block described above plus int32 xpending at word 12 and float sigmaPrev at word 13 (the library's own: sigma as
 * the alpha step saw it); 16 words, zero-initialised by the caller.
 * `own_off` / `n_own`: first owned cell and number of owned cells of the slab's grids (ghost planes excluded). */
int mf_cg_slab_after_dp(const double* gathered, int world, void* scalars, const int32_t* state_dev, int64_t own_off, int64_t n_own,
                        float* residual, float* tmp, double* maxabs_dev, int sx, int sy, int sz,
                        const int32_t* flags, const float* Aprecond, const float* Ai, const float* Aj, const float* Ak, double* dot_dev,
                        void* stream);
int mf_cg_slab_after_zr(const double* gathered, int world, void* scalars, float accuracy, int iter, int32_t* state_dev, int64_t own_off,
                        int64_t n_own, float* x, float* search, const float* tmp, void* stream);

/* ------------------------------------------------------------------------------------------------
 * HIP-only helpers (return an error in the CPU libraries)
 * ---------------------------------------------------------------------------------------------- */
/* average duration in microseconds of `reps` back-to-back launches of the named kernel on `stream`,
 * measured with hipEvents on that stream (used by bench.py for the roofline object).  mf_time_apply_matrix times the general
 * kernel (any matrix); _packed times the packed-coefficient variant the PCG loop runs after mf_mic_init on exactly these
 * flags / Ai / Aj / Ak when every off-diagonal is +0 or -1, and fails if those packed bytes do not exist. */
int mf_time_apply_matrix(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src,
                         const float* A0, const float* Ai, const float* Aj, const float* Ak, int reps,
                         double* avg_us_host, void* stream);
int mf_time_apply_matrix_packed(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src,
                                const float* A0, const float* Ai, const float* Aj, const float* Ak, int reps,
                                double* avg_us_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MANTA_HIP_H */
