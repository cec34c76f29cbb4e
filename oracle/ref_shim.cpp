/*
 * oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked or loaded by the product path).
 *
 * A C ABI over the reference's own C++ classes (zoharl3/mantaflow, compiled from /root/reference by
 * oracle/ref.mk in its NOPYTHON packaging).  This file is OUR code: it only constructs the reference's
 * FluidSolver / Grid / BasicParticleSystem objects around caller-owned arrays and calls the reference's
 * own plugins and kernels.  It pins the plain-C restatement (oracle/manta_oracle.c) and generates the
 * golden vectors under tests/golden/.
 *
 * Array conventions are those of include/manta_hip.h (Vec3/MAC grids and particle vectors are SoA here and
 * converted to the reference's AoS inside the shim).
 */
#include "manta.h"
#include "grid.h"
#include "particle.h"
#include "conjugategrad.h"
#include "commonkernels.h"
#include "levelset.h"
#include "mantaio.h"
#include "shapes.h"
#include "noisefield.h"
#include <cstring>
#include <string>
#include <vector>
#include <memory>

namespace Manta {
// free functions with external linkage defined in the reference's .cpp files (no header declares them)
void InitPreconditionModifiedIncompCholesky2(const FlagGrid& flags, Grid<Real>& Aprecond, Grid<Real>& A0,
                                             Grid<Real>& Ai, Grid<Real>& Aj, Grid<Real>& Ak);  // conjugategrad.cpp:66
void ApplyPreconditionModifiedIncompCholesky2(Grid<Real>& dst, Grid<Real>& Var1, const FlagGrid& flags,
                                              Grid<Real>& Aprecond, Grid<Real>& A0, Grid<Real>& Ai,
                                              Grid<Real>& Aj, Grid<Real>& Ak);                  // conjugategrad.cpp:135
// PYTHON() plugins (plain functions in the NOPYTHON packaging)
void advectSemiLagrange(const FlagGrid* flags, const MACGrid* vel, GridBase* grid, int order, Real strength,
                        int orderSpace, bool openBounds, int boundaryWidth, int clampMode, int orderTrace);  // advection.cpp:443
void computePressureRhs(Grid<Real>& rhs, const MACGrid& vel, const Grid<Real>& pressure, const FlagGrid& flags,
                        Real cgAccuracy, const Grid<Real>* phi, const Grid<Real>* perCellCorr,
                        const MACGrid* fractions, const MACGrid* obvel, Real gfClamp, Real cgMaxIterFac,
                        bool precondition, int preconditioner, bool enforceCompatibility, bool useL2Norm,
                        bool zeroPressureFixing, const Grid<Real>* curv, const Real surfTens);  // pressure.cpp:277
void solvePressure(MACGrid& vel, Grid<Real>& pressure, const FlagGrid& flags, Real cgAccuracy,
                   const Grid<Real>* phi, const Grid<Real>* perCellCorr, const MACGrid* fractions,
                   const MACGrid* obvel, Real gfClamp, Real cgMaxIterFac, bool precondition, int preconditioner,
                   bool enforceCompatibility, bool useL2Norm, bool zeroPressureFixing, const Grid<Real>* curv,
                   const Real surfTens, Grid<Real>* retRhs);                                     // pressure.cpp:482
void correctVelocity(MACGrid& vel, Grid<Real>& pressure, const FlagGrid& flags, Real cgAccuracy,
                     const Grid<Real>* phi, const Grid<Real>* perCellCorr, const MACGrid* fractions, Real gfClamp,
                     Real cgMaxIterFac, bool precondition, int preconditioner, bool enforceCompatibility,
                     bool useL2Norm, bool zeroPressureFixing, const Grid<Real>* curv, const Real surfTens);  // pressure.cpp:457
void mapPartsToMAC(const FlagGrid& flags, MACGrid& vel, MACGrid& velOld, const BasicParticleSystem& parts,
                   const ParticleDataImpl<Vec3>& partVel, Grid<Vec3>* weight, const ParticleDataImpl<int>* ptype,
                   const int exclude);                                                           // flip.cpp:637
void mapMACToParts(const FlagGrid& flags, const MACGrid& vel, const BasicParticleSystem& parts,
                   ParticleDataImpl<Vec3>& partVel, const ParticleDataImpl<int>* ptype, const int exclude);  // flip.cpp:717
void flipVelocityUpdate(const FlagGrid& flags, const MACGrid& vel, const MACGrid& velOld,
                        const BasicParticleSystem& parts, ParticleDataImpl<Vec3>& partVel, const Real flipRatio,
                        const ParticleDataImpl<int>* ptype, const int exclude);                  // flip.cpp:738
void mapPartsToGrid(const FlagGrid& flags, Grid<Real>& target, const BasicParticleSystem& parts,
                    const ParticleDataImpl<Real>& source);                                       // flip.cpp:682
void mapPartsToGridVec3(const FlagGrid& flags, Grid<Vec3>& target, const BasicParticleSystem& parts,
                        const ParticleDataImpl<Vec3>& source);                                   // flip.cpp:685
void mapGridToParts(const Grid<Real>& source, const BasicParticleSystem& parts, ParticleDataImpl<Real>& target);  // flip.cpp:699
void mapGridToPartsVec3(const Grid<Vec3>& source, const BasicParticleSystem& parts,
                        ParticleDataImpl<Vec3>& target);                                         // flip.cpp:702
void setWallBcs(const FlagGrid& flags, MACGrid& vel, const MACGrid* obvel, const MACGrid* fractions,
                const Grid<Real>* phiObs, int boundaryWidth);                                    // extforces.cpp:327
void addBuoyancy(const FlagGrid& flags, const Grid<Real>& density, MACGrid& vel, Vec3 gravity, Real coefficient,
                 bool scale);                                                                    // extforces.cpp:84
void addGravity(const FlagGrid& flags, MACGrid& vel, Vec3 gravity, const Grid<Real>* exclude, bool scale);  // extforces.cpp:62
void extrapolateMACSimple(FlagGrid& flags, MACGrid& vel, int distance, LevelsetGrid* phiObs, bool intoObs);  // fastmarch.cpp:337
void extrapolateMACFromWeight(MACGrid& vel, Grid<Vec3>& weight, int distance);                           // fastmarch.cpp:415
void markFluidCells(const BasicParticleSystem& parts, FlagGrid& flags, const Grid<Real>* phiObs,
                    const ParticleDataImpl<int>* ptype, const int exclude);                              // flip.cpp:166
void sampleFlagsWithParticles(const FlagGrid& flags, BasicParticleSystem& parts, const int discretization,
                              const Real randomness);                                                   // flip.cpp:33
void pushOutofObs(BasicParticleSystem& parts, const FlagGrid& flags, const Grid<Real>& phiObs, const Real shift,
                  const Real thresh, const ParticleDataImpl<int>* ptype, const int exclude);               // flip.cpp:598
void gridParticleIndex(const BasicParticleSystem& parts, ParticleIndexSystem& indexSys, const FlagGrid& flags,
                       Grid<int>& index, Grid<int>* counter);                                             // flip.cpp:273
void unionParticleLevelset(const BasicParticleSystem& parts, const ParticleIndexSystem& indexSys, const FlagGrid& flags,
                           const Grid<int>& index, LevelsetGrid& phi, const Real radiusFactor,
                           const ParticleDataImpl<int>* ptype, const int exclude);                        // flip.cpp:356
void extrapolateLsSimple(Grid<Real>& phi, int distance, bool inside, bool include_walls);                 // fastmarch.cpp:472
void setPartType(const BasicParticleSystem& parts, ParticleDataImpl<int>& ptype, const int mark, const int stype,
                 const FlagGrid& flags, const int cflag);                                                 // ptsplugins.cpp:62
void markIsolatedFluidCell(FlagGrid& flags, const int mark);                                              // grid.cpp:1008
void addForcePvel(ParticleDataImpl<Vec3>& vel, const Vec3& a, const Real dt, const ParticleDataImpl<int>* ptype,
                  const int exclude);                                                                     // ptsplugins.cpp:26
void updateVelocityFromDeltaPos(const BasicParticleSystem& parts, ParticleDataImpl<Vec3>& vel,
                                const ParticleDataImpl<Vec3>& x_prev, const Real dt, const ParticleDataImpl<int>* ptype,
                                const int exclude);                                                       // ptsplugins.cpp:38
void eulerStep(BasicParticleSystem& parts, const ParticleDataImpl<Vec3>& vel, const ParticleDataImpl<int>* ptype,
               const int exclude);                                                                        // ptsplugins.cpp:50
void densityInflow(const FlagGrid& flags, Grid<Real>& density, const WaveletNoiseField& noise, Shape* shape, Real scale,
                   Real sigma);                                                                           // initplugins.cpp:39
void computeEnergy(const FlagGrid& flags, const MACGrid& vel, Grid<Real>& energy);                           // waveletturbulence.cpp:191
void computeWaveletCoeffs(Grid<Real>& input);                                                                // :197
void applyNoiseVec3(const FlagGrid& flags, Grid<Vec3>& target, const WaveletNoiseField& noise, Real scale, Real scaleSpatial,
                    const Grid<Real>* weight, const Grid<Vec3>* uv);                                         // :156
void vorticityConfinement(MACGrid& vel, const FlagGrid& flags, Real strength, const Grid<Real>* strengthCell);  // extforces.cpp:419
void cgSolveDiffusion(const FlagGrid& flags, GridBase& grid, Real alpha, Real cgMaxIterFac, Real cgAccuracy);   // conjugategrad.cpp:350
void setOpenBound(FlagGrid& flags, int bWidth, std::string openBound, int type);
void resetOutflow(FlagGrid& flags, Grid<Real>* phi, BasicParticleSystem* parts, Grid<Real>* real, Grid<int>* index,
                  ParticleIndexSystem* indexSys);                                                                // extforces.cpp:134                             // extforces.cpp:106
void apicMapPartsToMAC(const FlagGrid& flags, MACGrid& vel, const BasicParticleSystem& parts, const ParticleDataImpl<Vec3>& partVel,
                       const ParticleDataImpl<Vec3>& cpx, const ParticleDataImpl<Vec3>& cpy, const ParticleDataImpl<Vec3>& cpz,
                       MACGrid* mass, const ParticleDataImpl<int>* ptype, const int exclude);                      // apic.cpp:92
void apicMapMACGridToParts(ParticleDataImpl<Vec3>& partVel, ParticleDataImpl<Vec3>& cpx, ParticleDataImpl<Vec3>& cpy,
                           ParticleDataImpl<Vec3>& cpz, const BasicParticleSystem& parts, const MACGrid& vel, const FlagGrid& flags,
                           const ParticleDataImpl<int>* ptype, const int exclude);                                // apic.cpp:175
void interpolateGrid(Grid<Real>& target, const Grid<Real>& source, Vec3 scale, Vec3 offset, Vec3i size, int orderSpace);      // waveletturbulence.cpp:37
void interpolateGridVec3(Grid<Vec3>& target, const Grid<Vec3>& source, Vec3 scale, Vec3 offset, Vec3i size, int orderSpace);  // :51
void interpolateMACGrid(MACGrid& target, const MACGrid& source, Vec3 scale, Vec3 offset, Vec3i size, int orderSpace);         // :73
}  // namespace Manta

using namespace Manta;

static thread_local std::string g_err;
#define SHIM_TRY try {
#define SHIM_CATCH                  \
	}                               \
	catch (std::exception & e) {    \
		g_err = e.what();           \
		return 1;                   \
	}                               \
	return 0;

namespace {
struct Ctx {
	FluidSolver solver;
	int64_t n;
	Ctx(int sx, int sy, int sz, float dt) : solver(Vec3i(sx, sy, sz), sz > 1 ? 3 : 2), n((int64_t)sx * sy * sz) {
		solver.mDt = dt;
	}
};
// reference AoS Vec3 grid filled from / written back to a SoA float[3][n] array
template <class G>
struct VecIO {
	G g;
	float* soa;
	int64_t n;
	bool wb;
	VecIO(Ctx& c, const float* s, bool writeback) : g(&c.solver), soa(const_cast<float*>(s)), n(c.n), wb(writeback) {
		if (soa)
			for (int64_t i = 0; i < n; i++) g[i] = Vec3(soa[i], soa[n + i], soa[2 * n + i]);
	}
	void store() {
		if (soa)
			for (int64_t i = 0; i < n; i++) {
				soa[i] = g[i].x;
				soa[n + i] = g[i].y;
				soa[2 * n + i] = g[i].z;
			}
	}
	~VecIO() {
		if (wb) store();
	}
	G* ptr() { return soa ? &g : nullptr; }
};
typedef VecIO<MACGrid> MacIO;
typedef VecIO<Grid<Vec3> > Vec3IO;

struct RealRef {  // zero-copy Real grid over caller memory (Grid(FluidSolver*, T* data) ctor, grid.cpp:63-73)
	std::unique_ptr<Grid<Real> > g;
	RealRef(Ctx& c, const float* p) {
		if (p) g.reset(new Grid<Real>(&c.solver, const_cast<float*>(p)));
	}
	Grid<Real>* ptr() { return g.get(); }
	Grid<Real>& ref() { return *g; }
};
struct LevelRef {
	std::unique_ptr<LevelsetGrid> g;
	LevelRef(Ctx& c, const float* p) {
		if (p) g.reset(new LevelsetGrid(&c.solver, const_cast<float*>(p)));
	}
};
struct Parts {
	BasicParticleSystem sys;
	int64_t np, stride;
	float* pos;
	int32_t* pflag;
	Parts(Ctx& c, int64_t np_, int64_t stride_, const float* pos_, const int32_t* pflag_)
	    : sys(&c.solver), np(np_), stride(stride_), pos(const_cast<float*>(pos_)), pflag(const_cast<int32_t*>(pflag_)) {
		sys.resizeAll(np);
		for (int64_t i = 0; i < np; i++) {
			sys[i].pos = Vec3(pos[i], pos[stride + i], pos[2 * stride + i]);
			sys[i].flag = pflag ? pflag[i] : 0;
		}
	}
	void store() {
		for (int64_t i = 0; i < np; i++) {
			pos[i] = sys[i].pos.x;
			pos[stride + i] = sys[i].pos.y;
			pos[2 * stride + i] = sys[i].pos.z;
			if (pflag) pflag[i] = sys[i].flag;
		}
	}
};
template <class T>
struct Pdata {
	ParticleDataImpl<T> pd;
	Pdata(Ctx& c, Parts& p) : pd(&c.solver) {
		p.sys.registerPdata(&pd);
		pd.resize(p.np);
	}
};
static void loadVec3(ParticleDataImpl<Vec3>& pd, const float* s, int64_t np, int64_t stride) {
	for (int64_t i = 0; i < np; i++) pd[i] = Vec3(s[i], s[stride + i], s[2 * stride + i]);
}
static void storeVec3(ParticleDataImpl<Vec3>& pd, float* s, int64_t np, int64_t stride) {
	for (int64_t i = 0; i < np; i++) {
		s[i] = pd[i].x;
		s[stride + i] = pd[i].y;
		s[2 * stride + i] = pd[i].z;
	}
}
}  // namespace

template <class G, class T>
static void uni_io(Ctx& c, int kind_write, const char* name, void* data, int64_t nelem) {
	G g(&c.solver);
	if (kind_write) {
		memcpy(&g[0], data, sizeof(T) * nelem);
		g.save(name);
	} else {
		g.load(name);
		memcpy(data, &g[0], sizeof(T) * nelem);
	}
}

extern "C" {

const char* mf_last_error(void) { return g_err.c_str(); }
const char* mf_backend(void) { return "reference"; }

/* ApplyMatrix / ApplyMatrix2D, conjugategrad.h:118-151 */
int ref_apply_matrix(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src, const float* A0,
                     const float* Ai, const float* Aj, const float* Ak) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	RealRef d(c, dst), s(c, src), a0(c, A0), ai(c, Ai), aj(c, Aj), ak(c, Ak);
	if (sz > 1)
		ApplyMatrix(fl, d.ref(), s.ref(), a0.ref(), ai.ref(), aj.ref(), ak.ref());
	else
		ApplyMatrix2D(fl, d.ref(), s.ref(), a0.ref(), ai.ref(), aj.ref(), ak.ref());
	SHIM_CATCH
}

/* MakeLaplaceMatrix, conjugategrad.h:154-187 */
int ref_make_laplace_matrix(int sx, int sy, int sz, const int32_t* flags, float* A0, float* Ai, float* Aj, float* Ak,
                            const float* fractions) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	RealRef a0(c, A0), ai(c, Ai), aj(c, Aj), ak(c, Ak);
	MacIO fr(c, fractions, false);
	MakeLaplaceMatrix(fl, a0.ref(), ai.ref(), aj.ref(), ak.ref(), fr.ptr());
	SHIM_CATCH
}

/* InitPreconditionModifiedIncompCholesky2, conjugategrad.cpp:66-97 */
int ref_mic_init(int sx, int sy, int sz, const int32_t* flags, float* Aprecond, const float* A0, const float* Ai,
                 const float* Aj, const float* Ak) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	RealRef ap(c, Aprecond), a0(c, A0), ai(c, Ai), aj(c, Aj), ak(c, Ak);
	InitPreconditionModifiedIncompCholesky2(fl, ap.ref(), a0.ref(), ai.ref(), aj.ref(), ak.ref());
	SHIM_CATCH
}

/* ApplyPreconditionModifiedIncompCholesky2, conjugategrad.cpp:135-159 */
int ref_mic_apply(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* var1, const float* Aprecond,
                  const float* A0, const float* Ai, const float* Aj, const float* Ak) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	RealRef d(c, dst), v(c, var1), ap(c, Aprecond), a0(c, A0), ai(c, Ai), aj(c, Aj), ak(c, Ak);
	ApplyPreconditionModifiedIncompCholesky2(d.ref(), v.ref(), fl, ap.ref(), a0.ref(), ai.ref(), aj.ref(), ak.ref());
	SHIM_CATCH
}

/* GridCg<ApplyMatrix>, conjugategrad.cpp:198-326, driven like solvePressureSystem (pressure.cpp:396-442) */
int ref_cg_solve(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* rhs, float* residual,
                 float* search, float* tmp, const float* A0, const float* Ai, const float* Aj, const float* Ak,
                 float* Aprecond, int pc, float accuracy, int maxIter, int useL2Norm, float* out) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	RealRef d(c, dst), r(c, rhs), res(c, residual), se(c, search), t(c, tmp), a0(c, A0), ai(c, Ai), aj(c, Aj), ak(c, Ak);
	RealRef ap(c, Aprecond);
	Grid<Real> p1(&c.solver), p2(&c.solver), p3(&c.solver);
	std::unique_ptr<GridCgInterface> gcg;
	if (sz > 1)
		gcg.reset(new GridCg<ApplyMatrix>(d.ref(), r.ref(), res.ref(), se.ref(), fl, t.ref(), a0.ptr(), ai.ptr(), aj.ptr(), ak.ptr()));
	else
		gcg.reset(new GridCg<ApplyMatrix2D>(d.ref(), r.ref(), res.ref(), se.ref(), fl, t.ref(), a0.ptr(), ai.ptr(), aj.ptr(), ak.ptr()));
	gcg->setAccuracy(accuracy);
	gcg->setUseL2Norm(useL2Norm != 0);
	if (pc == 2) gcg->setICPreconditioner(GridCgInterface::PC_mICP, ap.ptr(), &p1, &p2, &p3);
	for (int iter = 0; iter < maxIter; iter++)
		if (!gcg->iterate()) iter = maxIter;
	out[0] = gcg->getIterations();
	out[1] = gcg->getResNorm();
	out[2] = gcg->getSigma();
	SHIM_CATCH
}

/* computePressureRhs, pressure.cpp:277-299 */
int ref_compute_pressure_rhs(int sx, int sy, int sz, const int32_t* flags, float* rhs, const float* vel,
                             const float* phi, const float* perCellCorr, const float* fractions, const float* obvel,
                             float gfClamp, int enforceCompatibility, const float* curv, float surfTens) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	RealRef r(c, rhs), ph(c, phi), pcc(c, perCellCorr), cv(c, curv);
	Grid<Real> pressure(&c.solver);
	MacIO v(c, vel, false), fr(c, fractions, false), ov(c, obvel, false);
	computePressureRhs(r.ref(), v.g, pressure, fl, 1e-3, ph.ptr(), pcc.ptr(), fr.ptr(), ov.ptr(), gfClamp, 1.5, true, 1,
	                   enforceCompatibility != 0, false, false, cv.ptr(), surfTens);
	SHIM_CATCH
}

/* solvePressure, pressure.cpp:482-523 */
int ref_solve_pressure(int sx, int sy, int sz, float* vel, float* pressure, const int32_t* flags, float cgAccuracy,
                       const float* phi, const float* perCellCorr, const float* fractions, const float* obvel,
                       float gfClamp, float cgMaxIterFac, int precondition, int preconditioner,
                       int enforceCompatibility, int useL2Norm, int zeroPressureFixing, const float* curv,
                       float surfTens, float* retRhs) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	RealRef p(c, pressure), ph(c, phi), pcc(c, perCellCorr), cv(c, curv), rr(c, retRhs);
	MacIO v(c, vel, true), fr(c, fractions, false), ov(c, obvel, false);
	solvePressure(v.g, p.ref(), fl, cgAccuracy, ph.ptr(), pcc.ptr(), fr.ptr(), ov.ptr(), gfClamp, cgMaxIterFac,
	              precondition != 0, preconditioner, enforceCompatibility != 0, useL2Norm != 0, zeroPressureFixing != 0,
	              cv.ptr(), surfTens, rr.ptr());
	SHIM_CATCH
}

/* correctVelocity, pressure.cpp:457-478 */
int ref_correct_velocity(int sx, int sy, int sz, float* vel, const float* pressure, const int32_t* flags,
                         const float* phi, float gfClamp, const float* curv, float surfTens) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	RealRef p(c, pressure), ph(c, phi), cv(c, curv);
	MacIO v(c, vel, true);
	correctVelocity(v.g, p.ref(), fl, 1e-3, ph.ptr(), nullptr, nullptr, gfClamp, 1.5, true, 1, false, false, false,
	                cv.ptr(), surfTens);
	SHIM_CATCH
}

/* advectSemiLagrange, advection.cpp:443-461.  kind: 0 Real, 1 Vec3 (centred), 2 MAC, 3 Levelset(Real) */
int ref_advect_semi_lagrange(int sx, int sy, int sz, float dt, const int32_t* flags, const float* vel, float* grid,
                             int kind, int order, float strength, int orderSpace, int clampMode, int orderTrace) {
	SHIM_TRY
	Ctx c(sx, sy, sz, dt);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	MacIO v(c, vel, false);
	if (kind == 0) {
		// the plugin swaps data pointers with a pool grid (grid.cpp:100-111): use a pool grid and copy back
		Grid<Real> g(&c.solver);
		memcpy(&g[0], grid, sizeof(float) * c.n);
		advectSemiLagrange(&fl, &v.g, &g, order, strength, orderSpace, false, -1, clampMode, orderTrace);
		memcpy(grid, &g[0], sizeof(float) * c.n);
	} else if (kind == 3) {
		LevelsetGrid g(&c.solver);
		memcpy(&g[0], grid, sizeof(float) * c.n);
		advectSemiLagrange(&fl, &v.g, &g, order, strength, orderSpace, false, -1, clampMode, orderTrace);
		memcpy(grid, &g[0], sizeof(float) * c.n);
	} else if (kind == 1) {
		Vec3IO g(c, grid, false);
		advectSemiLagrange(&fl, &v.g, &g.g, order, strength, orderSpace, false, -1, clampMode, orderTrace);
		g.store();
	} else {
		MacIO g(c, grid, false);
		advectSemiLagrange(&fl, &v.g, &g.g, order, strength, orderSpace, false, -1, clampMode, orderTrace);
		g.store();
	}
	SHIM_CATCH
}

/* mapPartsToMAC, flip.cpp:637-661 */
int ref_map_parts_to_mac(int sx, int sy, int sz, const int32_t* flags, float* vel, float* velOld, float* weight,
                         int64_t np, int64_t pstride, const float* pos, const int32_t* pflag, const float* pvel,
                         const int32_t* ptype, int exclude) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	MacIO v(c, vel, true), vo(c, velOld, true);
	Vec3IO w(c, weight, true);
	Parts P(c, np, pstride, pos, pflag);
	Pdata<Vec3> pv(c, P);
	loadVec3(pv.pd, pvel, np, pstride);
	std::unique_ptr<Pdata<int> > pt;
	if (ptype) {
		pt.reset(new Pdata<int>(c, P));
		for (int64_t i = 0; i < np; i++) pt->pd[i] = ptype[i];
	}
	mapPartsToMAC(fl, v.g, vo.g, P.sys, pv.pd, w.ptr(), pt ? &pt->pd : nullptr, exclude);
	SHIM_CATCH
}

/* mapMACToParts, flip.cpp:717-721 */
int ref_map_mac_to_parts(int sx, int sy, int sz, const int32_t* flags, const float* vel, int64_t np, int64_t pstride,
                         const float* pos, const int32_t* pflag, float* pvel, const int32_t* ptype, int exclude) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	MacIO v(c, vel, false);
	Parts P(c, np, pstride, pos, pflag);
	Pdata<Vec3> pv(c, P);
	loadVec3(pv.pd, pvel, np, pstride);
	std::unique_ptr<Pdata<int> > pt;
	if (ptype) {
		pt.reset(new Pdata<int>(c, P));
		for (int64_t i = 0; i < np; i++) pt->pd[i] = ptype[i];
	}
	mapMACToParts(fl, v.g, P.sys, pv.pd, pt ? &pt->pd : nullptr, exclude);
	storeVec3(pv.pd, pvel, np, pstride);
	SHIM_CATCH
}

/* apicMapPartsToMAC, apic.cpp:92-110 */
int ref_apic_map_parts_to_mac(int sx, int sy, int sz, const int32_t* flags, float* vel, float* mass, int64_t np, int64_t pstride,
                              const float* pos, const int32_t* pflag, const float* pvel, const float* cpx, const float* cpy,
                              const float* cpz, const int32_t* ptype, int exclude) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	MacIO v(c, vel, true), m(c, mass, true);
	Parts P(c, np, pstride, pos, pflag);
	Pdata<Vec3> pv(c, P), px(c, P), py(c, P), pz(c, P);
	loadVec3(pv.pd, pvel, np, pstride);
	loadVec3(px.pd, cpx, np, pstride);
	loadVec3(py.pd, cpy, np, pstride);
	loadVec3(pz.pd, cpz, np, pstride);
	std::unique_ptr<Pdata<int> > pt;
	if (ptype) {
		pt.reset(new Pdata<int>(c, P));
		for (int64_t i = 0; i < np; i++) pt->pd[i] = ptype[i];
	}
	apicMapPartsToMAC(fl, v.g, P.sys, pv.pd, px.pd, py.pd, pz.pd, &m.g, pt ? &pt->pd : nullptr, exclude);
	SHIM_CATCH
}
/* apicMapMACGridToParts, apic.cpp:175-181 */
int ref_apic_map_mac_to_parts(int sx, int sy, int sz, const int32_t* flags, const float* vel, int64_t np, int64_t pstride,
                              const float* pos, const int32_t* pflag, float* pvel, float* cpx, float* cpy, float* cpz,
                              const int32_t* ptype, int exclude) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	MacIO v(c, vel, false);
	Parts P(c, np, pstride, pos, pflag);
	Pdata<Vec3> pv(c, P), px(c, P), py(c, P), pz(c, P);
	loadVec3(pv.pd, pvel, np, pstride);
	loadVec3(px.pd, cpx, np, pstride);
	loadVec3(py.pd, cpy, np, pstride);
	loadVec3(pz.pd, cpz, np, pstride);
	std::unique_ptr<Pdata<int> > pt;
	if (ptype) {
		pt.reset(new Pdata<int>(c, P));
		for (int64_t i = 0; i < np; i++) pt->pd[i] = ptype[i];
	}
	apicMapMACGridToParts(pv.pd, px.pd, py.pd, pz.pd, P.sys, v.g, fl, pt ? &pt->pd : nullptr, exclude);
	storeVec3(pv.pd, pvel, np, pstride);
	storeVec3(px.pd, cpx, np, pstride);
	storeVec3(py.pd, cpy, np, pstride);
	storeVec3(pz.pd, cpz, np, pstride);
	SHIM_CATCH
}

/* flipVelocityUpdate, flip.cpp:738-742 */
int ref_flip_velocity_update(int sx, int sy, int sz, const int32_t* flags, const float* vel, const float* velOld,
                             int64_t np, int64_t pstride, const float* pos, const int32_t* pflag, float* pvel,
                             float flipRatio, const int32_t* ptype, int exclude) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	MacIO v(c, vel, false), vo(c, velOld, false);
	Parts P(c, np, pstride, pos, pflag);
	Pdata<Vec3> pv(c, P);
	loadVec3(pv.pd, pvel, np, pstride);
	std::unique_ptr<Pdata<int> > pt;
	if (ptype) {
		pt.reset(new Pdata<int>(c, P));
		for (int64_t i = 0; i < np; i++) pt->pd[i] = ptype[i];
	}
	flipVelocityUpdate(fl, v.g, vo.g, P.sys, pv.pd, flipRatio, pt ? &pt->pd : nullptr, exclude);
	storeVec3(pv.pd, pvel, np, pstride);
	SHIM_CATCH
}

/* mapPartsToGrid / mapPartsToGridVec3, flip.cpp:682-687 */
int ref_map_parts_to_grid(int sx, int sy, int sz, int ncomp, const int32_t* flags, float* target, int64_t np,
                          int64_t pstride, const float* pos, const int32_t* pflag, const float* psrc) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	Parts P(c, np, pstride, pos, pflag);
	if (ncomp == 1) {
		RealRef t(c, target);
		Pdata<Real> ps(c, P);
		for (int64_t i = 0; i < np; i++) ps.pd[i] = psrc[i];
		mapPartsToGrid(fl, t.ref(), P.sys, ps.pd);
	} else {
		Vec3IO t(c, target, true);
		Pdata<Vec3> ps(c, P);
		loadVec3(ps.pd, psrc, np, pstride);
		mapPartsToGridVec3(fl, t.g, P.sys, ps.pd);
	}
	SHIM_CATCH
}

/* mapGridToParts / mapGridToPartsVec3, flip.cpp:699-704 */
int ref_map_grid_to_parts(int sx, int sy, int sz, int ncomp, const float* source, int64_t np, int64_t pstride,
                          const float* pos, const int32_t* pflag, float* ptarget) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	Parts P(c, np, pstride, pos, pflag);
	if (ncomp == 1) {
		RealRef s(c, source);
		Pdata<Real> pt(c, P);
		for (int64_t i = 0; i < np; i++) pt.pd[i] = ptarget[i];
		mapGridToParts(s.ref(), P.sys, pt.pd);
		for (int64_t i = 0; i < np; i++) ptarget[i] = pt.pd[i];
	} else {
		Vec3IO s(c, source, false);
		Pdata<Vec3> pt(c, P);
		loadVec3(pt.pd, ptarget, np, pstride);
		mapGridToPartsVec3(s.g, P.sys, pt.pd);
		storeVec3(pt.pd, ptarget, np, pstride);
	}
	SHIM_CATCH
}

/* ParticleSystem::advectInGrid, particle.h:526-550 */
int ref_advect_in_grid(int sx, int sy, int sz, const int32_t* flags, const float* vel, int64_t np, int64_t pstride,
                       float* pos, int32_t* pflag, float dt, int integrationMode, int deleteInObstacle,
                       int stopInObstacle, int skipNew, const int32_t* ptype, int exclude) {
	SHIM_TRY
	Ctx c(sx, sy, sz, dt);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	MacIO v(c, vel, false);
	Parts P(c, np, pstride, pos, pflag);
	std::unique_ptr<Pdata<int> > pt;
	if (ptype) {
		pt.reset(new Pdata<int>(c, P));
		for (int64_t i = 0; i < np; i++) pt->pd[i] = ptype[i];
	}
	P.sys.advectInGrid(fl, v.g, integrationMode, deleteInObstacle != 0, stopInObstacle != 0, skipNew != 0,
	                   pt ? &pt->pd : nullptr, exclude);
	P.store();
	SHIM_CATCH
}

/* setWallBcs (no fractions), extforces.cpp:327-335 */
int ref_set_wall_bcs(int sx, int sy, int sz, const int32_t* flags, float* vel, const float* obvel) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	MacIO v(c, vel, true), ov(c, obvel, false);
	setWallBcs(fl, v.g, ov.ptr(), nullptr, nullptr, 0);
	SHIM_CATCH
}

/* addBuoyancy, extforces.cpp:84-88 */
int ref_add_buoyancy(int sx, int sy, int sz, float dt, const int32_t* flags, const float* density, float* vel,
                     float gx, float gy, float gz, float coefficient, int scale) {
	SHIM_TRY
	Ctx c(sx, sy, sz, dt);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	RealRef d(c, density);
	MacIO v(c, vel, true);
	addBuoyancy(fl, d.ref(), v.g, Vec3(gx, gy, gz), coefficient, scale != 0);
	SHIM_CATCH
}

/* addGravity, extforces.cpp:62-66 */
int ref_add_gravity(int sx, int sy, int sz, float dt, const int32_t* flags, float* vel, float gx, float gy, float gz,
                    const float* exclude, int scale) {
	SHIM_TRY
	Ctx c(sx, sy, sz, dt);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	RealRef e(c, exclude);
	MacIO v(c, vel, true);
	addGravity(fl, v.g, Vec3(gx, gy, gz), e.ptr(), scale != 0);
	SHIM_CATCH
}

/* extrapolateMACSimple, fastmarch.cpp:337-376 */
int ref_extrapolate_mac_simple(int sx, int sy, int sz, int32_t* flags, float* vel, int distance, int intoObs) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, flags);
	MacIO v(c, vel, true);
	extrapolateMACSimple(fl, v.g, distance, nullptr, intoObs != 0);
	SHIM_CATCH
}
/* extrapolateMACFromWeight, fastmarch.cpp:415-430 */
int ref_extrapolate_mac_from_weight(int sx, int sy, int sz, float* vel, float* weight, int distance) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	MacIO v(c, vel, true);
	Vec3IO w(c, weight, true);
	extrapolateMACFromWeight(v.g, w.g, distance);
	SHIM_CATCH
}
/* markFluidCells, flip.cpp:166-188 */
int ref_mark_fluid_cells(int sx, int sy, int sz, int32_t* flags, int64_t np, int64_t pstride, const float* pos,
                         const int32_t* pflag, const int32_t* ptype, int exclude, const float* phiObs) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver);   // pool grid: the plugin swaps data pointers when phiObs is given (grid.cpp:99-111)
	memcpy(&fl[0], flags, sizeof(int32_t) * c.n);
	RealRef ph(c, phiObs);
	Parts P(c, np, pstride, pos, pflag);
	std::unique_ptr<Pdata<int> > pt;
	if (ptype) {
		pt.reset(new Pdata<int>(c, P));
		for (int64_t i = 0; i < np; i++) pt->pd[i] = ptype[i];
	}
	markFluidCells(P.sys, fl, ph.ptr(), pt ? &pt->pd : nullptr, exclude);
	memcpy(flags, &fl[0], sizeof(int32_t) * c.n);
	SHIM_CATCH
}
/* sampleFlagsWithParticles, flip.cpp:33-58: returns the particle count; pos (SoA, stride cap) filled up to cap */
int ref_sample_flags_with_particles(int sx, int sy, int sz, const int32_t* flags, int discretization, float randomness,
                                    int64_t cap, float* pos, int64_t* count) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	BasicParticleSystem sys(&c.solver);
	sampleFlagsWithParticles(fl, sys, discretization, randomness);
	*count = sys.size();
	for (int64_t i = 0; i < sys.size() && i < cap; i++) {
		pos[i] = sys[i].pos.x;
		pos[cap + i] = sys[i].pos.y;
		pos[2 * cap + i] = sys[i].pos.z;
	}
	SHIM_CATCH
}

struct PtypeIO {   // optional int pdata filled from / written back to a caller array
	std::unique_ptr<Pdata<int> > pt;
	int32_t* src;
	int64_t np;
	PtypeIO(Ctx& c, Parts& P, const int32_t* ptype) : src(const_cast<int32_t*>(ptype)), np(P.np) {
		if (ptype) {
			pt.reset(new Pdata<int>(c, P));
			for (int64_t i = 0; i < np; i++) pt->pd[i] = ptype[i];
		}
	}
	ParticleDataImpl<int>* ptr() { return pt ? &pt->pd : nullptr; }
	void store() {
		if (pt)
			for (int64_t i = 0; i < np; i++) src[i] = pt->pd[i];
	}
};
/* ParticleSystem::projectOutOfBnd, particle.h:592-604 */
int ref_project_out_of_bnd(int sx, int sy, int sz, int64_t np, int64_t pstride, float* pos, const int32_t* pflag,
                           float bnd, const char* plane, const int32_t* ptype, int exclude) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	std::vector<int> fdata(c.n, 0);
	FlagGrid fl(&c.solver, fdata.data());
	Parts P(c, np, pstride, pos, pflag);
	PtypeIO pt(c, P, ptype);
	P.sys.projectOutOfBnd(fl, bnd, std::string(plane), pt.ptr(), exclude);
	P.store();
	SHIM_CATCH
}
/* pushOutofObs, flip.cpp:584-602 */
int ref_push_out_of_obs(int sx, int sy, int sz, int64_t np, int64_t pstride, float* pos, const int32_t* pflag,
                        const float* phiObs, float shift, float thresh, const int32_t* ptype, int exclude) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	std::vector<int> fdata(c.n, 0);
	FlagGrid fl(&c.solver, fdata.data());
	RealRef ph(c, phiObs);
	Parts P(c, np, pstride, pos, pflag);
	PtypeIO pt(c, P, ptype);
	pushOutofObs(P.sys, fl, ph.ref(), shift, thresh, pt.ptr(), exclude);
	P.store();
	SHIM_CATCH
}
/* gridParticleIndex, flip.cpp:273-320: indexSys (capacity np) receives sourceIndex, *n_indexed its size */
int ref_grid_particle_index(int sx, int sy, int sz, int64_t np, int64_t pstride, const float* pos, const int32_t* pflag,
                            int32_t* indexSys, int32_t* index, int64_t* n_indexed) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	std::vector<int> fdata(c.n, 0);
	FlagGrid fl(&c.solver, fdata.data());
	Grid<int> idx(&c.solver, index);
	Parts P(c, np, pstride, pos, pflag);
	ParticleIndexSystem isys(&c.solver);
	gridParticleIndex(P.sys, isys, fl, idx, nullptr);
	*n_indexed = isys.size();
	for (int64_t i = 0; i < isys.size(); i++) indexSys[i] = isys[i].sourceIndex;
	SHIM_CATCH
}
/* gridParticleIndex + unionParticleLevelset, flip.cpp:322-363 (the index system is rebuilt here with the reference's own
 * gridParticleIndex so that the call takes plain arrays) */
int ref_union_particle_levelset(int sx, int sy, int sz, int64_t np, int64_t pstride, const float* pos, const int32_t* pflag,
                                float* phi, float radiusFactor, const int32_t* ptype, int exclude) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	std::vector<int> fdata(c.n, 0);
	FlagGrid fl(&c.solver, fdata.data());
	Grid<int> idx(&c.solver);
	Parts P(c, np, pstride, pos, pflag);
	PtypeIO pt(c, P, ptype);
	ParticleIndexSystem isys(&c.solver);
	gridParticleIndex(P.sys, isys, fl, idx, nullptr);
	LevelRef ph(c, phi);
	unionParticleLevelset(P.sys, isys, fl, idx, *ph.g, radiusFactor, pt.ptr(), exclude);
	SHIM_CATCH
}
/* extrapolateLsSimple, fastmarch.cpp:472-522 */
int ref_extrapolate_ls_simple(int sx, int sy, int sz, float* phi, int distance, int inside, int include_walls) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	RealRef ph(c, phi);
	extrapolateLsSimple(ph.ref(), distance, inside != 0, include_walls != 0);
	SHIM_CATCH
}
/* setPartType, ptsplugins.cpp:56-65 */
int ref_set_part_type(int sx, int sy, int sz, const int32_t* flags, int64_t np, int64_t pstride, const float* pos,
                      int32_t* ptype, int mark, int stype, int cflag) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	Parts P(c, np, pstride, pos, nullptr);
	PtypeIO pt(c, P, ptype);
	setPartType(P.sys, *pt.ptr(), mark, stype, fl, cflag);
	pt.store();
	SHIM_CATCH
}
/* markIsolatedFluidCell, grid.cpp:987-1011 */
int ref_mark_isolated_fluid_cell(int sx, int sy, int sz, int32_t* flags, int mark) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, flags);
	markIsolatedFluidCell(fl, mark);
	SHIM_CATCH
}
/* addForcePvel / updateVelocityFromDeltaPos / eulerStep, ptsplugins.cpp:20-53 */
int ref_add_force_pvel(int64_t np, int64_t pstride, float* pvel, float ax, float ay, float az, float dt,
                       const int32_t* ptype, int exclude) {
	SHIM_TRY
	Ctx c(4, 4, 4, dt);
	std::vector<float> zero(3 * (size_t)pstride, 0.f);
	Parts P(c, np, pstride, zero.data(), nullptr);
	PtypeIO pt(c, P, ptype);
	Pdata<Vec3> v(c, P);
	loadVec3(v.pd, pvel, np, pstride);
	addForcePvel(v.pd, Vec3(ax, ay, az), dt, pt.ptr(), exclude);
	storeVec3(v.pd, pvel, np, pstride);
	SHIM_CATCH
}
int ref_update_velocity_from_delta_pos(int64_t np, int64_t pstride, const float* pos, float* pvel, const float* xprev,
                                       float dt, const int32_t* ptype, int exclude) {
	SHIM_TRY
	Ctx c(4, 4, 4, dt);
	Parts P(c, np, pstride, pos, nullptr);
	PtypeIO pt(c, P, ptype);
	Pdata<Vec3> v(c, P), xp(c, P);
	loadVec3(v.pd, pvel, np, pstride);
	loadVec3(xp.pd, xprev, np, pstride);
	updateVelocityFromDeltaPos(P.sys, v.pd, xp.pd, dt, pt.ptr(), exclude);
	storeVec3(v.pd, pvel, np, pstride);
	SHIM_CATCH
}
int ref_euler_step(int64_t np, int64_t pstride, float* pos, const float* pvel, float dt, const int32_t* ptype, int exclude) {
	SHIM_TRY
	Ctx c(4, 4, 4, dt);
	Parts P(c, np, pstride, pos, nullptr);
	PtypeIO pt(c, P, ptype);
	Pdata<Vec3> v(c, P);
	loadVec3(v.pd, pvel, np, pstride);
	eulerStep(P.sys, v.pd, pt.ptr(), exclude);
	P.store();
	SHIM_CATCH
}
/* LevelsetGrid::join / subtract, levelset.cpp:107-118; Grid::setBound, grid.cpp:629-637 */
int ref_levelset_join(int64_t n, float* phi, const float* other) {
	SHIM_TRY
	Ctx c((int)n, 1, 1, 1.f);
	LevelRef a(c, phi), b(c, other);
	a.g->join(*b.g);
	SHIM_CATCH
}
int ref_levelset_subtract(int64_t n, float* phi, const float* other, const int32_t* flags, int subtractType) {
	SHIM_TRY
	Ctx c((int)n, 1, 1, 1.f);
	LevelRef a(c, phi), b(c, other);
	std::unique_ptr<FlagGrid> fl;
	if (flags) fl.reset(new FlagGrid(&c.solver, const_cast<int*>(flags)));
	a.g->subtract(*b.g, fl.get(), subtractType);
	SHIM_CATCH
}
/* resetOutflow, extforces.cpp:134-161.  The reference compacts the particle array afterwards; the survivors are returned
 * in order (pos SoA [3][pstride], first *np_out entries), so the caller can compare them with its non-deleted particles. */
int ref_reset_outflow(int sx, int sy, int sz, int32_t* flags, float* phi, float* real, int64_t np, int64_t pstride, float* pos,
                      int32_t* pflag, int64_t* np_out) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, flags);
	RealRef p(c, phi), r(c, real);
	std::unique_ptr<Parts> P;
	if (pos) P.reset(new Parts(c, np, pstride, pos, pflag));
	resetOutflow(fl, p.ptr(), P ? &P->sys : nullptr, r.ptr(), nullptr, nullptr);
	if (P) {
		const int64_t m = P->sys.size();
		for (int64_t i = 0; i < m; i++) {
			pos[i] = P->sys[i].pos.x;
			pos[pstride + i] = P->sys[i].pos.y;
			pos[2 * pstride + i] = P->sys[i].pos.z;
			pflag[i] = P->sys[i].flag;
		}
		*np_out = m;
	}
	SHIM_CATCH
}
/* cgSolveDiffusion, conjugategrad.cpp:350-423.  kind 1: Real grid [n]; 3: MAC grid SoA [3][n] */
int ref_cg_solve_diffusion(int sx, int sy, int sz, const int32_t* flags, float* grid, int kind, float alpha, float cgMaxIterFac,
                           float cgAccuracy) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	if (kind == 1) {
		RealRef g(c, grid);
		cgSolveDiffusion(fl, g.ref(), alpha, cgMaxIterFac, cgAccuracy);
	} else {
		MacIO v(c, grid, true);
		cgSolveDiffusion(fl, v.g, alpha, cgMaxIterFac, cgAccuracy);
	}
	SHIM_CATCH
}
int ref_grid_set_bound(int sx, int sy, int sz, float* grid, float value, int w) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	RealRef g(c, grid);
	g.ref().setBound(value, w);
	SHIM_CATCH
}

/* Grid<T>::save / load through the reference's own .uni / .raw writers and readers (fileio/iogrids.cpp).
 * kind: 0 int, 1 Real, 2 Vec3 (AoS float[n][3]), 3 MAC, 4 Levelset */
int ref_grid_file(int sx, int sy, int sz, int kind, int write, const char* name, void* data) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	switch (kind) {
		case 0: uni_io<Grid<int>, int>(c, write, name, data, c.n); break;
		case 1: uni_io<Grid<Real>, Real>(c, write, name, data, c.n); break;
		case 2: uni_io<Grid<Vec3>, Vec3>(c, write, name, data, c.n); break;
		case 3: uni_io<MACGrid, Vec3>(c, write, name, data, c.n); break;
		default: uni_io<LevelsetGrid, Real>(c, write, name, data, c.n); break;
	}
	SHIM_CATCH
}

/* interpolateGrid / interpolateGridVec3 / interpolateMACGrid, plugin/waveletturbulence.cpp:37-78.  kind 0 Real, 1 Vec3, 2 MAC */
int ref_interpolate_grid(int kind, int tsx, int tsy, int tsz, float* target, int ssx, int ssy, int ssz, const float* source,
                         float scx, float scy, float scz, float ox, float oy, float oz, int zx, int zy, int zz, int orderSpace) {
	SHIM_TRY
	Ctx ct(tsx, tsy, tsz, 1.f), cs(ssx, ssy, ssz, 1.f);
	const Vec3 scale(scx, scy, scz), off(ox, oy, oz);
	const Vec3i size(zx, zy, zz);
	if (kind == 0) {
		RealRef t(ct, target), s(cs, source);
		interpolateGrid(t.ref(), s.ref(), scale, off, size, orderSpace);
	} else if (kind == 1) {
		Vec3IO t(ct, target, true), s(cs, source, false);
		interpolateGridVec3(t.g, s.g, scale, off, size, orderSpace);
	} else {
		MacIO t(ct, target, true), s(cs, source, false);
		interpolateMACGrid(t.g, s.g, scale, off, size, orderSpace);
	}
	SHIM_CATCH
}

/* shapes: kind 0 Box(p0 = a, p1 = b), 1 Sphere(center a, radius b.x, scale c), 2 Cylinder(center a, radius b.x, z c) */
static Shape* make_shape(Ctx& c, int kind, const float* q) {
	const Vec3 a(q[0], q[1], q[2]), b(q[3], q[4], q[5]), cc(q[6], q[7], q[8]);
	if (kind == 0) return new Box(&c.solver, Vec3::Invalid, a, b, Vec3::Invalid);
	if (kind == 1) return new Sphere(&c.solver, a, b.x, cc);
	return new Cylinder(&c.solver, a, b.x, cc);
}
/* Shape::computeLevelset, shapes.cpp */
int ref_shape_levelset(int sx, int sy, int sz, int kind, const float* q, float* phi) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	std::unique_ptr<Shape> sh(make_shape(c, kind, q));
	LevelRef ph(c, phi);
	sh->generateLevelset(*ph.g);
	SHIM_CATCH
}
/* Shape::applyToGrid needs the Python argument store (NOPYTHON build: errMsg), so the kernels ApplyShapeToGrid<T> /
 * ApplyShapeToMACGrid (shapes.cpp:40-69) are applied here with the shape's own isInside / isInsideGrid.
 * gridkind 0 Real [n], 1 Vec3 SoA [3][n], 2 MAC SoA [3][n], 3 int [n] */
int ref_shape_apply(int sx, int sy, int sz, int kind, const float* q, int gridkind, void* grid, const float* value,
                    const int32_t* respectFlags) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	std::unique_ptr<Shape> sh(make_shape(c, kind, q));
	const int64_t n = c.n;
	float* gf = (float*)grid;
	int32_t* gi = (int32_t*)grid;
	std::unique_ptr<FlagGrid> fl;
	if (respectFlags) fl.reset(new FlagGrid(&c.solver, const_cast<int*>(respectFlags)));
	Grid<Real> dummy(&c.solver);
	FOR_IJK(dummy) {
		const int64_t idx = dummy.index(i, j, k);
		if (fl && fl->isObstacle(i, j, k)) continue;
		if (gridkind == 2) {
			if (sh->isInside(Vec3(i, j + 0.5, k + 0.5))) gf[idx] = value[0];
			if (sh->isInside(Vec3(i + 0.5, j, k + 0.5))) gf[n + idx] = value[1];
			if (sh->isInside(Vec3(i + 0.5, j + 0.5, k))) gf[2 * n + idx] = value[2];
		} else if (sh->isInsideGrid(i, j, k)) {
			if (gridkind == 0) gf[idx] = value[0];
			else if (gridkind == 3) gi[idx] = (int32_t)value[0];
			else {
				gf[idx] = value[0];
				gf[n + idx] = value[1];
				gf[2 * n + idx] = value[2];
			}
		}
	}
	SHIM_CATCH
}
/* the reference's wavelet noise tile (3 x 128^3, generated once per process) and seed offset */
int ref_noise_tile(float* out) {
	SHIM_TRY
	Ctx c(8, 8, 8, 1.f);
	WaveletNoiseField nf(&c.solver, -1, 0);
	memcpy(out, nf.data(), sizeof(float) * 3 * 128 * 128 * 128);
	SHIM_CATCH
}
/* densityInflow, initplugins.cpp:27-43: P = posScale[3], posOffset[3], valOffset, valScale, clamp, clampNeg, clampPos, timeAnim */
int ref_density_inflow(int sx, int sy, int sz, float timeTotal, const int32_t* flags, float* density, int kind, const float* q,
                       int fixedSeed, const float* P, float scale, float sigma) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	c.solver.mTimeTotal = timeTotal;
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	RealRef d(c, density);
	std::unique_ptr<Shape> sh(make_shape(c, kind, q));
	WaveletNoiseField nf(&c.solver, fixedSeed, 0);
	nf.mPosScale = Vec3(P[0], P[1], P[2]);
	nf.mPosOffset = Vec3(P[3], P[4], P[5]);
	nf.mValOffset = P[6];
	nf.mValScale = P[7];
	nf.mClamp = P[8] != 0.f;
	nf.mClampNeg = P[9];
	nf.mClampPos = P[10];
	nf.mTimeAnim = P[11];
	densityInflow(fl, d.ref(), nf, sh.get(), scale, sigma);
	SHIM_CATCH
}

/* scenes/simpleplume.py driven through the reference's own classes and plugins (the loop below is this shim's code: same
 * calls, same arguments, same order as the scene's main loop), `steps` steps at resolution `res`.  Outputs: density
 * [sz][sy][sx], velocity SoA [3][sz][sy][sx]. */
int ref_simpleplume(int res, int steps, int inflow_steps, float* density_out, float* vel_out) {
	SHIM_TRY
	const Vec3i gsi(res, int(1.5 * res), res);
	const Vec3 gs(gsi.x, gsi.y, gsi.z);
	FluidSolver solver(gsi, 3);
	FlagGrid flags(&solver);
	MACGrid vel(&solver);
	Grid<Real> density(&solver), pressure(&solver);
	WaveletNoiseField noise(&solver, -1, 0);
	noise.mPosScale = Vec3(45);
	noise.mClamp = true;
	noise.mClampNeg = 0;
	noise.mClampPos = 1;
	noise.mValOffset = 0.75;
	noise.mTimeAnim = 0.2;
	Cylinder source(&solver, gs * Vec3(0.5, 0.1, 0.5), res * 0.14, gs * Vec3(0, 0.02, 0));
	flags.initDomain(0, "xXyYzZ", "      ", "      ", "      ", nullptr);
	flags.fillGrid();
	for (int t = 0; t < steps; t++) {
		if (t < inflow_steps) densityInflow(flags, density, noise, &source, 1, 0.5);
		advectSemiLagrange(&flags, &vel, &density, 2, 1.0, 1, false, -1, 2, 1);
		advectSemiLagrange(&flags, &vel, &vel, 2, 1.0, 1, false, -1, 2, 1);
		setWallBcs(flags, vel, nullptr, nullptr, nullptr, 0);
		addBuoyancy(flags, density, vel, Vec3(0, -6e-4, 0), 1.0, true);
		solvePressure(vel, pressure, flags, 1e-3, nullptr, nullptr, nullptr, nullptr, 0.01, 1.5, true, 1, false, false, false,
		              nullptr, 0., nullptr);
		solver.step();
	}
	const int64_t n = (int64_t)gsi.x * gsi.y * gsi.z;
	for (int64_t i = 0; i < n; i++) {
		density_out[i] = density[i];
		vel_out[i] = vel[i].x;
		vel_out[n + i] = vel[i].y;
		vel_out[2 * n + i] = vel[i].z;
	}
	SHIM_CATCH
}

/* scenes/waveletTurbulence.py driven through the reference's own classes and plugins (the loop is this shim's code: same calls, same
 * arguments, same order as the scene's main loop).  Shape::applyToGrid needs the Python argument store (NOPYTHON build: errMsg), so
 * the MAC branch is applied here with the shape's own isInside at the three face positions (shapes.cpp:61-68).
 * Outputs: coarse density/velocity and the up-sampled (xl) density/velocity, SoA. */
static void shim_apply_to_mac(MACGrid& g, Shape& sh, Vec3 value) {
	FOR_IJK(g) {
		if (sh.isInside(Vec3(i, j + 0.5, k + 0.5))) g(i, j, k).x = value.x;
		if (sh.isInside(Vec3(i + 0.5, j, k + 0.5))) g(i, j, k).y = value.y;
		if (sh.isInside(Vec3(i + 0.5, j + 0.5, k))) g(i, j, k).z = value.z;
	}
}
static void shim_wlt_noise(WaveletNoiseField& n, WaveletNoiseField* like, Real timeAnim) {
	n.mPosScale = like ? like->mPosScale : Vec3(20);
	n.mClamp = true;
	n.mClampNeg = 0;
	n.mClampPos = 2;
	n.mValScale = 1;
	n.mValOffset = 0.075;
	n.mTimeAnim = timeAnim;
}
int ref_waveletturbulence(int res, int dim, int steps, int upres, float* density_out, float* vel_out, float* xl_density_out,
                          float* xl_vel_out) {
	SHIM_TRY
	const double wltStrength = 0.4; // a Python float in the scene
	Vec3i gsi(res, int(1.5 * res), res);
	if (dim == 2) gsi.z = 1;
	const Vec3 gs(gsi.x, gsi.y, gsi.z);
	FluidSolver sm(gsi, dim);
	sm.mDt = 1.5;
	const Vec3 velInflow(0.025, 0, 0);
	WaveletNoiseField noise(&sm, 265, 0);
	shim_wlt_noise(noise, nullptr, 0.3);
	Cylinder source(&sm, gs * Vec3(0.3, 0.2, 0.5), res * 0.081, gs * Vec3(0.081, 0, 0));
	Cylinder sourceVel(&sm, gs * Vec3(0.3, 0.2, 0.5), res * 0.15, gs * Vec3(0.15, 0, 0));
	Vec3i xlgsi(upres * gsi.x, upres * gsi.y, upres * gsi.z);
	if (dim == 2) xlgsi.z = 1;
	const Vec3 xl_gs(xlgsi.x, xlgsi.y, xlgsi.z);
	FluidSolver xl(xlgsi, dim);
	xl.mDt = sm.mDt;
	FlagGrid xl_flags(&xl);
	MACGrid xl_vel(&xl);
	Grid<Real> xl_density(&xl), xl_weight(&xl);
	xl_flags.initDomain(0, "xXyYzZ", "      ", "      ", "      ", nullptr);
	xl_flags.fillGrid();
	Cylinder xl_source(&xl, xl_gs * Vec3(0.3, 0.2, 0.5), xl_gs.x * 0.081, xl_gs * Vec3(0.081, 0, 0));
	WaveletNoiseField xl_noise(&xl, 265, 0);
	shim_wlt_noise(xl_noise, &noise, 0.3 * upres);
	WaveletNoiseField wlt1(&xl, -1, 0), wlt2(&xl, -1, 0), wlt3(&xl, -1, 0);
	wlt1.mPosScale = Vec3(int(1.0 * gs.x)) * 0.5;
	wlt1.mTimeAnim = 0.1;
	wlt2.mPosScale = wlt1.mPosScale * 2.0;
	wlt2.mTimeAnim = 0.1;
	wlt3.mPosScale = wlt2.mPosScale * 2.0;
	wlt3.mTimeAnim = 0.1;
	FlagGrid flags(&sm);
	MACGrid vel(&sm);
	Grid<Real> density(&sm), pressure(&sm), energy(&sm);
	const int bWidth = 0;
	flags.initDomain(bWidth, "xXyYzZ", "      ", "      ", "      ", nullptr);
	flags.fillGrid();
	setOpenBound(flags, bWidth, "Y", FlagGrid::TypeOutflow | FlagGrid::TypeEmpty);
	for (int t = 0; t < steps; t++) {
		advectSemiLagrange(&flags, &vel, &density, 2, 1.0, 1, false, -1, 2, 1);
		advectSemiLagrange(&flags, &vel, &vel, 2, 1.0, 1, false, -1, 2, 1);
		bool applyInflow = false;
		if (sm.getTime() >= 0 && sm.getTime() < 50.) {
			densityInflow(flags, density, noise, &source, 1, 0.5);
			shim_apply_to_mac(vel, sourceVel, velInflow * float(res));
			applyInflow = true;
		}
		setWallBcs(flags, vel, nullptr, nullptr, nullptr, 0);
		addBuoyancy(flags, density, vel, Vec3(0, -1e-3, 0), 1.0, true);
		vorticityConfinement(vel, flags, 0.3, nullptr);
		solvePressure(vel, pressure, flags, 0.01, nullptr, nullptr, nullptr, nullptr, 0.01, 1.0, true, 1, false, false, false, nullptr,
		              0., nullptr);
		setWallBcs(flags, vel, nullptr, nullptr, nullptr, 0);
		computeEnergy(flags, vel, energy);
		computeWaveletCoeffs(energy);
		sm.step();
		interpolateGrid(xl_weight, energy, Vec3(1.), Vec3(0.), Vec3i(-1, -1, -1), 1);
		interpolateMACGrid(xl_vel, vel, Vec3(1.), Vec3(0.), Vec3i(-1, -1, -1), 1);
		applyNoiseVec3(xl_flags, xl_vel, wlt1, wltStrength * 1.0, 1.0, &xl_weight, nullptr);
		applyNoiseVec3(xl_flags, xl_vel, wlt2, wltStrength * 0.6, 1.0, &xl_weight, nullptr);
		applyNoiseVec3(xl_flags, xl_vel, wlt3, wltStrength * 0.6 * 0.6, 1.0, &xl_weight, nullptr);
		for (int sub = 0; sub < upres; sub++) advectSemiLagrange(&xl_flags, &xl_vel, &xl_density, 2, 1.0, 1, false, -1, 2, 1);
		if (applyInflow) densityInflow(xl_flags, xl_density, xl_noise, &xl_source, 1, 0.5);
		xl.step();
	}
	const int64_t n = (int64_t)gsi.x * gsi.y * gsi.z, nx = (int64_t)xlgsi.x * xlgsi.y * xlgsi.z;
	for (int64_t i = 0; i < n; i++) {
		density_out[i] = density[i];
		vel_out[i] = vel[i].x;
		vel_out[n + i] = vel[i].y;
		vel_out[2 * n + i] = vel[i].z;
	}
	for (int64_t i = 0; i < nx; i++) {
		xl_density_out[i] = xl_density[i];
		xl_vel_out[i] = xl_vel[i].x;
		xl_vel_out[nx + i] = xl_vel[i].y;
		xl_vel_out[2 * nx + i] = xl_vel[i].z;
	}
	SHIM_CATCH
}

/* computeEnergy / computeWaveletCoeffs / vorticityConfinement / setOpenBound / applyNoiseVec3 */
int ref_compute_energy(int sx, int sy, int sz, const int32_t* flags, const float* vel, float* energy) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	MacIO v(c, vel, false);
	RealRef e(c, energy);
	computeEnergy(fl, v.g, e.ref());
	SHIM_CATCH
}
int ref_compute_wavelet_coeffs(int sx, int sy, int sz, float* input) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	RealRef e(c, input);
	computeWaveletCoeffs(e.ref());
	SHIM_CATCH
}
int ref_vorticity_confinement(int sx, int sy, int sz, float* vel, const int32_t* flags, float strength, const float* strengthCell) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	MacIO v(c, vel, true);
	RealRef sc(c, strengthCell);
	vorticityConfinement(v.g, fl, strength, sc.ptr());
	SHIM_CATCH
}
int ref_set_open_bound(int sx, int sy, int sz, int32_t* flags, int bWidth, const char* openBound, int type) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, flags);
	setOpenBound(fl, bWidth, std::string(openBound), type);
	SHIM_CATCH
}
/* P as in ref_density_inflow; weight grid of size (wsx, wsy, wsz), nullable */
int ref_apply_noise_vec3(int sx, int sy, int sz, float timeTotal, const int32_t* flags, float* target, int fixedSeed, const float* P,
                         float scale, float scaleSpatial, const float* weight, int wsx, int wsy, int wsz, const float* uv) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	c.solver.mTimeTotal = timeTotal;
	FlagGrid fl(&c.solver, const_cast<int*>(flags));
	Vec3IO t(c, target, true);
	WaveletNoiseField nf(&c.solver, fixedSeed, 0);
	nf.mPosScale = Vec3(P[0], P[1], P[2]);
	nf.mPosOffset = Vec3(P[3], P[4], P[5]);
	nf.mValOffset = P[6];
	nf.mValScale = P[7];
	nf.mClamp = P[8] != 0.f;
	nf.mClampNeg = P[9];
	nf.mClampPos = P[10];
	nf.mTimeAnim = P[11];
	std::unique_ptr<Ctx> cw;
	std::unique_ptr<RealRef> w;
	std::unique_ptr<Vec3IO> u;      /* the uv grid has the size (wsx, wsy, wsz) as well (the plugin asserts it) */
	if (weight || uv) cw.reset(new Ctx(wsx, wsy, wsz, 1.f));
	if (weight) w.reset(new RealRef(*cw, weight));
	if (uv) u.reset(new Vec3IO(*cw, uv, false));
	applyNoiseVec3(fl, t.g, nf, scale, scaleSpatial, w ? w->ptr() : nullptr, u ? &u->g : nullptr);
	SHIM_CATCH
}

/* Grid<Real>::getMaxAbs (grid.cpp:356-360) and GridSumSqr (commonkernels.h:32-35) */
int ref_grid_max_abs(int64_t n, const float* a, float* out) {
	SHIM_TRY
	Ctx c((int)n, 1, 1, 1.f);
	RealRef g(c, a);
	*out = g.ref().getMaxAbs();
	SHIM_CATCH
}
int ref_grid_sum_sqr(int64_t n, const float* a, double* out) {
	SHIM_TRY
	Ctx c((int)n, 1, 1, 1.f);
	RealRef g(c, a);
	*out = GridSumSqr(g.ref()).sum;
	SHIM_CATCH
}

/* FlagGrid::initDomain + fillGrid (grid.cpp:798-927) -- reference flag patterns for fixtures */
int ref_init_domain(int sx, int sy, int sz, int32_t* flags, int boundaryWidth, const char* wall, const char* open,
                    const char* inflow, const char* outflow, int fillType) {
	SHIM_TRY
	Ctx c(sx, sy, sz, 1.f);
	FlagGrid fl(&c.solver, flags);
	fl.initDomain(boundaryWidth, wall, open, inflow, outflow, nullptr);
	if (fillType) fl.fillGrid(fillType);
	SHIM_CATCH
}

}  // extern "C"
