// pressure.h -- what pressure.hip (PCG loop) and mic.hip (MIC(0) preconditioner sweeps) share.
#pragma once
#include "common.h"

// device-resident PCG scalars (GridCg members, conjugategrad.h:95-112): written by one-block kernels, read by every
// kernel of an iteration; `done` makes the kernels of iterations queued after convergence return at once
struct CgScalars {
	float sigma, alpha, nalpha, beta, resNorm, dp, sigmaNew, accuracy;
	int iterations, done, diverged, useL2;
	int xpending;     // this iteration's dst += alpha * search is still to be done (by k_cg_update_search_x)
	float sigmaPrev;  // z-slab solver: sigma as the alpha step saw it (the beta step's divisor while block 0 of the same kernel writes sigma)
};

// the beta step of the PCG (k_cg_beta: residual norm, convergence test, beta; conjugategrad.cpp:268-295) as the tail of the backward MIC
// sweep's last workgroup: the partials of the residual update (an earlier kernel) and the nsig dot partials (this launch's, plus what
// the caller appended behind them)
struct BetaTail {
	CgScalars* sc;            // nullptr: no tail
	int nbr;                  // blocks of the residual update
	const float* fpart;       // their min / max pairs
	const double* dpart_res;  // their sums of squares (L2 norm)
	int nsig;                 // dot partials to fold
	// sc == nullptr and sum_out set: the z-slab solver's variant -- no beta step here (the scalars of all ranks are gathered first), the
	// last workgroup only folds: *sum_out = sum of the nsig dot partials (k_mic_fin_sum's order) and, with nbr > 0, *maxabs_out =
	// max |fpart| unless live->done (k_fin_maxabs_live)
	double* sum_out;
	double* maxabs_out;
	const CgScalars* live;
};
namespace mf {
// mode 0: InitPreconditionModifiedIncompCholesky2 (dst := Aprecond, var1 := A0); 1 / 2: forward / backward substitution
// of ApplyPreconditionModifiedIncompCholesky2.  sc (nullable): skip when sc->done.
int mic_launch(int mode, const Dim& d, const int32_t* flags, float* dst, const float* var1, const float* Ap, const float* Ai,
               const float* Aj, const float* Ak, const CgScalars* sc, hipStream_t st);
// backward substitution with GridDotProduct(dst, var1) fused into the sweep ("rows" mode): *ndot partials in dotpart, summed in
// index order by the caller; *ndot == 0 when the active mode cannot fuse it (the caller then runs its own dot kernel)
int mic_launch_dot(const Dim& d, const int32_t* flags, float* dst, const float* var1, const float* Ap, const float* Ai,
                   const float* Aj, const float* Ak, const CgScalars* sc, double* dotpart, int* ndot, hipStream_t st, bool empty_ext = false,
                   BetaTail tail = BetaTail{nullptr, 0, nullptr, nullptr, 0}, bool* tail_done = nullptr);
// forward + backward substitution with dot(dst, var1) -> *dot_dev and (nbr > 0) the max |.| of the nbr min / max pairs in fpart ->
// *maxabs_dev (skipped once live->done): folded by the backward sweep's last workgroup where the active mode allows, by one-block
// kernels behind it otherwise
int mic_apply_dot_fold(const Dim& d, const int32_t* flags, float* dst, const float* var1, const float* Ap, const float* Ai,
                       const float* Aj, const float* Ak, double* dot_dev, int nbr, const float* fpart, double* maxabs_dev,
                       const CgScalars* live, hipStream_t st);
// empty_ext: the caller sums the shares of the bundles the sweep leaves out itself (mic_empty_map tells which), their entries come out 0
// (any_size: also where every workgroup draws one bundle only and the sweep sums the shares of the empty ones itself -- the map is then only
// what the vector kernels of the PCG skip by)
int mic_empty_map(const Dim& d, const int32_t* flags, const float* Ap, const float* Aj, const float* Ak, const int** bempty, int* nbj, hipStream_t st,
                  bool any_size = false);
// packed {fluid, Ai, Aj, Ak} bytes built by the last mf_mic_init for exactly these grids (nullptr when unavailable / not exact);
// synchronises the stream once.  *a0_packed: bits 4-7 of every byte hold the (small integer) diagonal A0 of these grids as well
int mic_pack_query(const Dim& d, const int32_t* flags, const float* A0, const float* Ai, const float* Aj, const float* Ak,
                   const unsigned char** pack, bool* a0_packed, hipStream_t st);
// packed bytes built by mf_pack_matrix for exactly these grids, or nullptr (no synchronisation); *a0_packed: they carry this A0
const unsigned char* mic_pack_user(const int32_t* flags, const float* A0, const float* Ai, const float* Aj, const float* Ak, bool* a0_packed);
int mic_mode();          // the requested sweep mode: 0 levels, 2 rows
// until reset with (0, 0): the apply sweeps of the registered system (on its packed bytes) cover the cells [xoff, xoff + 8 nchunks) of
// every row only.  The caller guarantees that every cell outside has a zero packed byte (non-fluid, no couplings) and the value +0 in
// the swept grid, and accounts for nothing of them in the fused dot (their products are +0).
void mic_set_trim(int xoff_cells, int nchunks);
// matrix-free set-up of mf_solve_pressure_fused: buffers of the system handle (empty-bundle map preset to 1 = empty), then the MIC
// factor from the packed bytes and the registration of (flags, Aprecond) as a system without coefficient arrays
int mic_fused_begin(const Dim& d, hipStream_t st, unsigned char** pack, int** bempty, int* nbj);
int mic_fused_finish(const Dim& d, const int32_t* flags, float* Aprecond, hipStream_t st);
int mic_flow_error();    // reads (and clears) the deadlock-guard flag of the single-launch sweeps; needs a synchronised stream
}  // namespace mf
