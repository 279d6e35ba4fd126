/*
 * bicgstab_hip.h -- C ABI of libbicgstab_hip.so: the MI355X (gfx950) BiCGStab hot path.
 *
 * Drop-in boundary: the four solver entry points of the reference's solver.h, byte-for-byte the
 * same signatures and struct layouts, so that the reference's C host (main.c + matrix.c + mmio.c)
 * links against this library instead of its own solver.c / vector.c.
 * Everything else in this header is additive (handle-based API for callers that keep the matrix
 * resident on the GPU, kernel-level entry points for parity tests and benchmarks, communicator
 * bootstrap). Plain C types only.
 *
 * All arithmetic is fp64, all indices uint32, exactly as in the reference (src/matrix.h:19-26).
 */
#ifndef BICGSTAB_HIP_H
#define BICGSTAB_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * Matrix containers. If the reference's matrix.h was included first (its include guard is
 * MATRIX_H) its own typedefs are used; otherwise layout-identical ones are declared here.
 *   CSR_Matrix  <- reference src/matrix.h:19-26   (sizeof 40)
 *   INFO_Matrix <- reference src/matrix.h:28-33   (sizeof 32; MM_typecode = char[4], mmio.h:16)
 * Per rank: diag = rows x rows block with LOCAL column indices, offd = rows x n block with
 * GLOBAL column indices (reference src/matrix.c:343-351, 380-392); displs[p]/recvcounts[p] =
 * first row / row count of rank p (src/matrix.c:306-307).
 * ------------------------------------------------------------------------------------------- */
#ifndef MATRIX_H
typedef struct {
    double       *val;
    unsigned int *col;
    unsigned int *ptr;
    unsigned int  nz;
    unsigned int  rows;
    unsigned int  cols;
} CSR_Matrix;

typedef struct {
    unsigned int nz, rows, cols;
    char         code[4];
    int         *recvcounts;
    int         *displs;
} INFO_Matrix;
#endif

/* ---------------------------------------------------------------------------------------------
 * 1. Drop-in entry points (replace reference src/solver.c).
 *
 *   bicgstab          <- src/solver.h:10 (implementation src/solver.c:35-146)
 *   ca_bicgstab       <- src/solver.h:11 (src/solver.c:160-278)
 *   pipe_bicgstab     <- src/solver.h:12 (src/solver.c:292-417)
 *   pipe_bicgstab_rr  <- src/solver.h:13 (src/solver.c:433-576)
 *
 * Semantics kept: collective over all ranks; x_loc in = x0, out = solution; r_loc in = b,
 * out = recursive residual (src/solver.c:75); return value = iterations executed; non-square
 * matrix prints "Error: matrix is not square." and exit(1) (src/solver.c:43-46); rank 0 prints
 * the reference's progress and summary lines verbatim (src/solver.c:124,135-139). The matrix is
 * never modified. Constants default to the reference's (EPS 1e-15, MAX_ITER 1000, OUT_ITER 100;
 * src/solver.c:3-9) and can be overridden without an ABI change through the environment:
 * BICG_TOL, BICG_MAX_ITER, BICG_OUT_ITER, BICG_CHECK_EVERY, BICG_QUIET, BICG_RR_DRIFT (adaptive
 * residual replacement for the pipelined solvers, off by default).
 *
 * Rank discovery: if the process has initialised MPI (the reference's main.c does) the library
 * picks rank/size up from MPI_COMM_WORLD through weak symbols and bootstraps RCCL with an
 * MPI_Bcast of the unique id; otherwise it runs single-rank unless a bicg_comm_init_* call was
 * made first. HIP / RCCL failures print to stderr and exit(EXIT_FAILURE).
 * ------------------------------------------------------------------------------------------- */
int bicgstab(CSR_Matrix *A_loc_diag, CSR_Matrix *A_loc_offd, INFO_Matrix *A_info, double *x_loc, double *r_loc);
int ca_bicgstab(CSR_Matrix *A_loc_diag, CSR_Matrix *A_loc_offd, INFO_Matrix *A_info, double *x_loc, double *r_loc);
int pipe_bicgstab(CSR_Matrix *A_loc_diag, CSR_Matrix *A_loc_offd, INFO_Matrix *A_info, double *x_loc, double *r_loc);
int pipe_bicgstab_rr(CSR_Matrix *A_loc_diag, CSR_Matrix *A_loc_offd, INFO_Matrix *A_info, double *x_loc, double *r_loc, int krr, int nrr);

/* Shifted systems (A + sigma_j I) x_j = b, j = 0..sigma_len-1, solved with ONE Krylov recurrence on
 * the seed system (2 SpMV per iteration whatever the number of shifts) -- all six entry points of
 * reference src/shifted_solver.h:16-21:
 *
 *   shifted_bicgstab                    <- :16 (src/shifted_solver.c:13-180)   seed = A, xi/tau recurrences
 *   shifted_lopbicgstab                 <- :17 (:182-354)   seed = A + sigma[seed] I
 *   shifted_lopbicgstab_v2              <- :18 (:357-529)   same arithmetic as :17, re-ordered
 *   shifted_lopbicgstab_nooverlap       <- :19 (:531-701)   same arithmetic as :17, re-ordered
 *   shifted_pipe_lopbicgstab            <- :20 (:703-895)   pipelined recurrence
 *   shifted_pipe_lopbicgstab_nooverlap  <- :21 (:897-1086)  same arithmetic as :20, re-ordered
 *
 * (the re-ordered reference functions are bit-identical to their base function; each group resolves
 * to one implementation here). x_loc_set is shift-major, x_loc_set[j * local_rows + i]
 * (src/shifted_solver.c:118-123), in = initial guess (the reference's drivers pass 0), out = the
 * sigma_len solutions; r_loc in = b, out = seed residual. Convergence: max_j |1/(zeta_j pi_j)|^2 (r,r)
 * <= EPS^2 (b,b) (xi/tau form: max_j |xi_j tau_j|), EPS 1e-12, MAX_ITER 1000 (src/shifted_solver.c:5-6). */
int shifted_bicgstab(CSR_Matrix *A_loc_diag, CSR_Matrix *A_loc_offd, INFO_Matrix *A_info, double *x_loc_set, double *r_loc, double *sigma, int sigma_len);
int shifted_lopbicgstab(CSR_Matrix *A_loc_diag, CSR_Matrix *A_loc_offd, INFO_Matrix *A_info, double *x_loc_set, double *r_loc, double *sigma, int sigma_len, int seed);
int shifted_lopbicgstab_v2(CSR_Matrix *A_loc_diag, CSR_Matrix *A_loc_offd, INFO_Matrix *A_info, double *x_loc_set, double *r_loc, double *sigma, int sigma_len, int seed);
int shifted_lopbicgstab_nooverlap(CSR_Matrix *A_loc_diag, CSR_Matrix *A_loc_offd, INFO_Matrix *A_info, double *x_loc_set, double *r_loc, double *sigma, int sigma_len, int seed);
int shifted_pipe_lopbicgstab(CSR_Matrix *A_loc_diag, CSR_Matrix *A_loc_offd, INFO_Matrix *A_info, double *x_loc_set, double *r_loc, double *sigma, int sigma_len, int seed);
int shifted_pipe_lopbicgstab_nooverlap(CSR_Matrix *A_loc_diag, CSR_Matrix *A_loc_offd, INFO_Matrix *A_info, double *x_loc_set, double *r_loc, double *sigma, int sigma_len, int seed);
/* Shifted solvers with per-shift convergence flags and SEED SWITCHING, reference
 * src/shifted_switching_solver.h:10-12:
 *   shifted_lopbicg                   <- :10 (src/shifted_switching_solver.c:20-257)   a shift whose residual bound
 *                                        |1/(zeta pi)| ||r|| has reached EPS is frozen; ends when all have converged
 *   shifted_lopbicg_switching         <- :11 (:260-608)   as above; when the seed converges first, the slowest
 *                                        remaining shift becomes the seed (history of alpha/beta/omega/pi re-derived)
 *   shifted_lopbicg_switching_noovlp  <- :12 (:611-1016)  same arithmetic, different MPI_Wait placement
 * Same arguments as shifted_lopbicgstab. r_loc returns the (rescaled) residual of the final seed. Return value as
 * in the reference: iterations for shifted_lopbicg, iterations + 1 for the switching variants (k starts at 1). */
int shifted_lopbicg(CSR_Matrix *A_loc_diag, CSR_Matrix *A_loc_offd, INFO_Matrix *A_info, double *x_loc_set, double *r_loc, double *sigma, int sigma_len, int seed);
int shifted_lopbicg_switching(CSR_Matrix *A_loc_diag, CSR_Matrix *A_loc_offd, INFO_Matrix *A_info, double *x_loc_set, double *r_loc, double *sigma, int sigma_len, int seed);
int shifted_lopbicg_switching_noovlp(CSR_Matrix *A_loc_diag, CSR_Matrix *A_loc_offd, INFO_Matrix *A_info, double *x_loc_set, double *r_loc, double *sigma, int sigma_len, int seed);

/* ---------------------------------------------------------------------------------------------
 * 2. Communicator bootstrap (process-global; one process per GPU).
 *    The reference uses MPI_COMM_WORLD implicitly (src/solver.c:37, src/matrix.c:432); here the
 *    transport is explicit: RCCL over xGMI for production, or a host-staged transport driven by
 *    caller-supplied callbacks (MPI, gloo, ...) for tests and for more ranks than GPUs.
 * ------------------------------------------------------------------------------------------- */
#define BICG_UNIQUE_ID_BYTES 128

/* rank 0: create an RCCL unique id to broadcast to the other ranks by any means */
int bicg_comm_unique_id(void *id_out /* BICG_UNIQUE_ID_BYTES */);
/* all ranks: join. device = HIP device ordinal for this process (-1: rank % device count). Returns 0; with
 * BICG_COMM_SOFT_FAIL=1 in the environment a failing ncclCommInitRank returns 1 instead of exiting (no
 * communicator is installed then -- the caller picks another transport on all ranks). */
int bicg_comm_init_rccl(int rank, int nranks, const void *id, int device);

/* Host-staged transport. Callbacks operate on HOST buffers and are collective.
 *   allreduce_sum: in-place sum of n doubles over all ranks (replaces MPI_Iallreduce+MPI_Wait of
 *                  one double each, e.g. reference src/solver.c:90-91, here packed)
 *   alltoallv:     byte-wise personalised exchange (counts/displs in BYTES, per peer), replaces
 *                  the full-vector MPI_Iallgatherv of reference src/matrix.c:432 with a halo */
typedef void (*bicg_allreduce_fn)(double *buf, int n, void *user);
typedef void (*bicg_alltoallv_fn)(const void *send, const int *scounts, const int *sdispls,
                                  void *recv, const int *rcounts, const int *rdispls, void *user);
int bicg_comm_init_host(int rank, int nranks, bicg_allreduce_fn allreduce, bicg_alltoallv_fn alltoallv,
                        void *user, int device);
/* MPI_COMM_WORLD of the calling process through weak symbols; transport "rccl" or "host".
 * Returns non-zero when the process has no initialised MPI. */
int bicg_comm_init_mpi(const char *transport, int device);
int bicg_comm_init_single(int device);
/* Collective, after one of the init calls above: switch the DATA path (halo values, dot sums) to direct
 * peer-to-peer stores between the GPUs of one node -- mailboxes mapped through HIP IPC, written by the
 * producing kernels, no library collective per exchange (the replacement for the MPI_Iallgatherv /
 * MPI_Iallreduce pair of reference src/matrix.c:432, src/solver.c:90 that a 20 us iteration can afford).
 * Ends with a self-test over the real links; returns 0 when it passed on EVERY rank, otherwise nothing
 * changes and the transport's own collectives stay in use. bicg_comm_init_mpi("auto") tries it itself.
 * BICG_P2P=0 disables it. */
int bicg_comm_enable_p2p(void);
/* 0: not active; 1: active, mailboxes in ordinary device memory; 2: active, uncached device memory */
int bicg_comm_p2p_active(void);
void bicg_comm_finalize(void);
/* 1 when librccl and every entry point the RCCL transport uses resolve (dies with a message otherwise); no device call */
int bicg_comm_rccl_loadable(void);
/* Why the last bicg_comm_init_rccl of this process failed: RCCL's result string and, where the library has it, its own last
 * error text ("" after a success). Returns the text's length. With BICG_COMM_SOFT_FAIL=1 a refused communicator is reported
 * this way instead of ending the process (bench.py records RCCL's refusal of two ranks on one device verbatim). */
int bicg_comm_last_error(char *out, int cap);
/* one-rank RCCL round trip (library load, communicator, all-reduce); 0 = ok. Needs a GPU. */
int bicg_comm_selftest_rccl(int device);
int bicg_comm_rank(void);
int bicg_comm_size(void);

/* ---------------------------------------------------------------------------------------------
 * 3. Handle-based API: upload once, solve many times, vectors may stay resident in HBM.
 * ------------------------------------------------------------------------------------------- */
typedef struct bicg_ctx bicg_ctx;

enum { BICG_BICGSTAB = 0, BICG_CA_BICGSTAB = 1, BICG_PIPE_BICGSTAB = 2, BICG_PIPE_BICGSTAB_RR = 3 };

typedef struct {
    double tol;          /* relative residual tolerance, reference EPS (src/solver.c:3) */
    int    max_iter;     /* reference MAX_ITER (src/solver.c:4) */
    int    out_iter;     /* progress line every out_iter iterations, reference OUT_ITER (:9); 0 = never */
    int    check_every;  /* iterations enqueued between host convergence checks (device stops itself) */
    int    quiet;        /* 1: no stdout lines */
    int    krr, nrr;     /* residual replacement period / count (src/solver.c:433) */
    int    record_trace; /* 1: keep per-iteration alpha/omega/beta/(r,r) on the device */
    int    time_kernels; /* bit 0: give every SpMV kernel its own start/stop HIP events (roofline measurement);
                            bit 1: section timing -- the reference's MEASURE_SECTION_TIME (src/shifted_switching_solver.c:9,
                            src/shifted_solver.c:77-81, 230-247) on the device clock: an event wherever the kind of work changes
                            (product / element-wise / shifted systems / reduction hand-over), read with bicg_section_times.
                            Both keep the multi-launch forms (no persistent launch, no graph replay);
                            bit 2 (with bit 1; BICG_SECTION_TIME=2): the switching solvers also print the reference's
                            DISPLAY_SECTION_TIME table, one line per iteration, and its ten totals incl. "Switch time"
                            (src/shifted_switching_solver.c:884-892, 994-1005) */
    double rr_drift;     /* pipelined solvers, additive: > 0 enables ADAPTIVE residual
                            replacement -- at every host check the true residual b - A x is computed and, when
                            ||(b - A x) - r|| > rr_drift * ||r||, the next iteration is a replacement step
                            (the step of src/solver.c:498-508, 522-531). 0 = the reference's fixed krr/nrr policy only. */
} bicg_options;

typedef struct {
    int    iterations;     /* k, the reference's return value */
    double dot_r;          /* final (r,r), all ranks */
    double dot_zero;       /* (r0,r0) */
    double seconds;        /* wall time, init SpMV .. last iteration (the span of src/solver.c:70-131) */
    double iter_seconds;   /* wall time of the iteration loop only (after the set-up phase) */
    double spmv_ms_total;  /* sum of the kernel durations of all timed SpMVs (time_kernels) */
    int    spmv_launches;  /* number of SpMVs timed (an SpMV may consist of up to 4 kernels) */
    int    breakdown_iteration; /* first iteration with a non-finite alpha/beta/omega/(r,r); 0 = none.
                              The reference does not detect breakdown (it iterates on NaNs). */
    int    adaptive_replacements; /* replacement steps triggered by rr_drift */
} bicg_result;

void bicg_default_options(bicg_options *o);

/* Upload this rank's blocks and build the SpMV plan (row blocks, halo lists). Collective.
 * Returns NULL after printing to stderr on failure. The host arrays are not referenced afterwards. */
bicg_ctx *bicg_create(const CSR_Matrix *diag, const CSR_Matrix *offd, const INFO_Matrix *info);
void bicg_destroy(bicg_ctx *ctx);
/* Single rank, matrix ALREADY in device memory as CSR (device pointers, 32-bit indices, ptr_d[rows] = nnz): the sliced-ELL plan
 * is built by kernels, no host copy of the matrix is made -- what lets BASELINE.json configs[3] at its stated size (512^3
 * Laplacian: 134 M rows, 938 M non-zeros) be generated, planned and solved on ONE MI355X inside a bench run. The arrays are
 * not referenced after the call. plan_seconds (optional): wall time of the call. Returns NULL (after a line on stderr) for
 * blocks that need the host plan: ragged rows (jagged slices / x windows), long rows (rows over lanes), more than one rank.
 *   bicg_stencil7_device  CSR of the 7-point stencil on an m^3 grid in device memory (weights = centre, x-, x+, y-, y+, z-, z+;
 *                         entries in ascending column order: the matrix of the reference-style generator used by bench.py and
 *                         the tests, python/synth.py stencil7); arrays are released with bicg_device_free */
bicg_ctx *bicg_create_device_csr(const double *val_d, const unsigned int *col_d, const unsigned int *ptr_d, unsigned int rows,
                                 double *plan_seconds);
int bicg_stencil7_device(unsigned int m, const double *weights, double **val_d, unsigned int **col_d, unsigned int **ptr_d,
                         unsigned long long *nnz);
void bicg_device_free(void *p);

/* Matrix residency of the drop-in entry points (section 1 and the shifted ones): the context of the last drop-in call
 * stays resident and is reused when the caller passes the same blocks again -- same arrays, sizes, partition,
 * communicator AND contents (a 64-bit hash over every value / column / row pointer: the matrix may have been edited in
 * place, e.g. by the reference's csr_shift_diagonal, src/matrix.c:518-531). The reference's drivers call a solver 10-28 x
 * on one matrix (src/main_repeat.c:109-132): only the first call pays for plan + upload. All ranks agree on hit / miss.
 * BICG_DROPIN_CACHE=0 in the environment restores create / destroy per call.
 *   bicg_dropin_context  the resident context for these blocks (created or reused; collective); owned by the library --
 *                        do NOT bicg_destroy it. Lets a host form b = A*1 with bicg_spmv and then call bicgstab() with
 *                        one upload (host/bicg_main.c)
 *   bicg_dropin_release  drop the resident context now (also happens in bicg_comm_finalize)
 *   bicg_dropin_stats    hits / misses so far */
bicg_ctx *bicg_dropin_context(const CSR_Matrix *diag, const CSR_Matrix *offd, const INFO_Matrix *info);
void bicg_dropin_release(void);
void bicg_dropin_stats(unsigned int *hits, unsigned int *misses);

/* Whole solve with host vectors, semantics of section 1 (x_loc/r_loc in-out). */
int bicg_solve(bicg_ctx *ctx, int method, double *x_loc, double *r_loc, const bicg_options *opt,
               bicg_result *res);

/* Device-resident variant: load x0/b once, run, fetch. bicg_run leaves x and r on the device. */
int bicg_load(bicg_ctx *ctx, const double *x0_loc, const double *b_loc);
int bicg_run(bicg_ctx *ctx, int method, const bicg_options *opt, bicg_result *res);
int bicg_fetch(bicg_ctx *ctx, double *x_loc, double *r_loc);
/* The same solve in three steps (benchmarks time exactly K iterations this way):
 *   begin   = the reference's set-up phase (src/solver.c:74-83, 200-213, 333-348), synchronised
 *   iterate = up to nsteps further passes of its while loop (stops early on convergence or
 *             max_iter); returns k so far; synchronised on return
 *   end     = summary lines (src/solver.c:134-141) and result */
int bicg_run_begin(bicg_ctx *ctx, int method, const bicg_options *opt);
int bicg_run_iterate(bicg_ctx *ctx, int nsteps);
/* bicg_run_iterate with three clocks around it (ms): [0] the device's -- an event recorded on the compute stream in front of the
 * first launch and one behind the last (hipEventElapsedTime); [1] the host time spent enqueueing the launches (until the loop over
 * the iterations returned, before the scalars are fetched); [2] the host's wall time of the whole call. A region whose wall time
 * is long while [0] is not was held up on the host side (or between the queue and the device), not in the kernels. */
int bicg_run_iterate_timed(bicg_ctx *ctx, int nsteps, double ms[3]);
int bicg_run_end(bicg_ctx *ctx, bicg_result *res);
int bicg_sync(bicg_ctx *ctx);
/* shifted solve on a resident matrix; variant = BICG_SHIFTED_LOP / _PIPE / _XI / _FLAG / _SWITCH (semantics
 * of the drop-in functions above; opt NULL = reference defaults with EPS 1e-12); the trace holds the SEED
 * system's alpha, omega, beta, (r,r). _FLAG / _SWITCH (shifted_lopbicg / shifted_lopbicg_switching): the
 * return value follows the reference (the switching variants count from 1), bicg_result.iterations is the
 * number of iterations and bicg_result.adaptive_replacements the number of seed switches. */
enum { BICG_SHIFTED_LOP = 0, BICG_SHIFTED_PIPE = 1, BICG_SHIFTED_XI = 2, BICG_SHIFTED_FLAG = 3, BICG_SHIFTED_SWITCH = 4 };
int bicg_solve_shifted(bicg_ctx *ctx, int variant, double *x_loc_set, double *r_loc, const double *sigma, int sigma_len,
                       int seed, const bicg_options *opt, bicg_result *res);
/* The check of the reference's shifted driver (src/test_shifted.c:129-154) on the device: relres_out[j] =
 * || (A + sigma_j I) x_j - b || / || b || for every shift (x_loc_set shift-major as above). Collective. */
int bicg_shifted_residuals(bicg_ctx *ctx, const double *x_loc_set, const double *b_loc, const double *sigma, int sigma_len,
                           double *relres_out);
/* "Batched SpMV" (BASELINE.json configs[4]): Y_j = (A + sigma_j I) X_j for nvec vectors with every matrix entry read once
 * per 16 vectors (sliced-ELL SpMM) -- what the loop above does internally. x_loc_set / y_loc_set are shift-major like
 * x_loc_set of the shifted solvers; sigma may be NULL (Y_j = A X_j). Every column is bit-identical to bicg_spmv of that
 * vector. Padded and jagged slices, x-window slots and lane-permuted groups are all read (BICG_FLAG_SPMM); returns 1 without
 * computing only when some rows are on the CSR / rows-over-lanes kernels (a row several times longer than the average). ms_device
 * (optional) receives the device time of the passes (halo exchanges, layout change and SpMM kernel). Collective. */
int bicg_spmm(bicg_ctx *ctx, const double *x_loc_set, const double *sigma, int nvec, double *y_loc_set, double *ms_device);
/* per-iteration trace of the last run (record_trace): arrays of length >= iterations, may be NULL */
int bicg_trace(bicg_ctx *ctx, double *alpha, double *omega, double *beta, double *dot_r);

/* ---------------------------------------------------------------------------------------------
 * 4. Kernel-level entry points (parity tests, benchmarks).
 *   bicg_spmv        <- MPI_csr_spmv_ovlap, reference src/matrix.c:428-441 (collective)
 *   bicg_dot         <- my_ddot + MPI_Iallreduce, reference src/vector.c:9-15, src/solver.c:89-91
 * ------------------------------------------------------------------------------------------- */
int bicg_spmv(bicg_ctx *ctx, const double *x_loc, double *y_loc);
double bicg_dot(bicg_ctx *ctx, const double *x_loc, const double *y_loc);
/* reps back-to-back SpMVs on device-resident vectors; returns average ms per SpMV (HIP events on
 * the library's compute stream) */
int bicg_spmv_bench(bicg_ctx *ctx, int reps, double *ms_per_spmv);
/* STREAM-style bandwidth of THIS GPU, the denominator north_star prices the SpMV against ("measure
 * its own STREAM-triad/copy on the box"): kind 0 copy a = b, 1 triad a = b + s c (both 16-byte accesses), 2 read-only with
 * 8-byte loads, 3 read-only with 16-byte loads. bytes_per_array should exceed the 256 MiB Infinity Cache several times
 * (bench.py: 1 GiB). gbps = bytes read + written per second; ms = one pass. Returns 0. Needs a GPU. */
int bicg_stream_bench(int kind, unsigned long long bytes_per_array, int reps, double *gbps, double *ms);
/* 1 after a peer-to-peer wait of this context timed out (only reachable with BICG_P2P_SOFT_FAIL=1; the
 * default is to print the error and exit like any other HIP/RCCL failure). The solve in progress stops. */
int bicg_comm_failed(bicg_ctx *ctx);
/* How long the exchanges of the last solve made the persistent kernels wait (multi-rank, peer-to-peer data path): one sample per
 * exchange, taken on the device clock inside the launch. out = {all-reduce of a dot group through the mailboxes: p50, p99; a
 * boundary workgroup's wait from publishing its own values to a complete window, the neighbour's halo values included: p50, p99
 * (microseconds); number of samples of either}. Returns 0 when something was recorded. What a first multi-GPU run needs to
 * explain itself: the links' latency as the kernels saw it (reference overlap: src/solver.c:363-367, src/matrix.c:432-440). */
int bicg_comm_wait_stats(bicg_ctx *ctx, double out[6]);
/* Section times of the last solve run with bicg_options.time_kernels & 2 (BICG_SECTION_TIME=1 in the drop-in path), in
 * milliseconds on the compute stream: ms[0] element-wise kernels of the seed system (with their fused dots), ms[1] products
 * A x (halo exchange and joins of overlapped all-reduces included), ms[2] the passes over the shifted systems (the shifted
 * solvers print "Seed time" / "Shift time" like the reference: shift = ms[2], seed = total - shift,
 * src/shifted_solver.c:230-247), ms[3] reduction hand-overs that are launches of their own (host / RCCL all-reduce + apply,
 * peer-to-peer collect); with one rank the scalars are applied inside the producing kernels and ms[3] is 0.
 * *iterations = iterations covered, *marks = events used (negative: the pool of 65536 ran out and timing stopped there).
 * Returns 0, or 2 when the last solve was not timed. */
int bicg_section_times(bicg_ctx *c, double ms[4], int *iterations, int *marks);

/* plan facts: local rows, diag nnz, offd nnz, halo length, workgroups per SpMV, halo-touching
 * workgroups, rows on the sliced-ELL path, sliced-ELL padding entries */
int bicg_plan_info(bicg_ctx *ctx, unsigned int out[8]);
/* which code paths this context takes (tests assert on them): peer-to-peer data path in use; halo exchange
 * folded into the sliced-ELL SpMV launch; two-stream overlap mode; 16-bit column offsets; every 256-row
 * group on the sliced-ELL path; jagged slices (ragged rows, no padding stored); bicg_spmm available (the same
 * on every rank); x windows in LDS with 16-bit slots instead of column indices; rows spread over lanes */
enum { BICG_FLAG_P2P = 1, BICG_FLAG_LL_FUSED = 2, BICG_FLAG_OVERLAP = 4, BICG_FLAG_COL16 = 8, BICG_FLAG_ALL_SELL = 16, BICG_FLAG_JAGGED = 32,
       BICG_FLAG_SPMM = 64, BICG_FLAG_WINDOW = 128,
       BICG_FLAG_ROWSPLIT = 256   /* long rows: a row is spread over 8..64 lanes (k_spmv_rows); row sums then agree with the
                                     reference's to 1e-13 x sum |a_ij x_j| instead of bit for bit */,
       BICG_FLAG_PERSIST = 512    /* pipe_bicgstab runs as ONE persistent launch per chunk of iterations (latency-bound ranks:
                                     matrix slices and x window in LDS, vectors in registers; bicg_persist.hip) */,
       BICG_FLAG_FUSE_PIPE = 1024 /* multi-launch pipelined iterations run their element-wise phases in the SpMV epilogues (two
                                     launches per iteration) rather than as separate kernels */,
       BICG_FLAG_PIPE_PROBED = 2048 /* ... and that was MEASURED on this matrix by the first pipelined solve (BICG_PLAN="pipe-probe")
                                     instead of decided by the size / layout rule of bicg_create */,
       BICG_FLAG_UNIFORM = 4096   /* some 64-row slices are UNIFORM -- all rows present, equally long, entry k at the same distance
                                     from its row in every row (the interior of a banded or stencil matrix): the SpMV takes their
                                     columns from one shared list of distances and reads no column index for them (8 instead of
                                     10 / 12 bytes per non-zero); bicg_uniform_entries counts those entries */,
       BICG_FLAG_CONSTANT = 8192  /* ... and some of them are CONSTANT: entry k also holds the same value in all 64 rows (the interior
                                     of a constant-coefficient stencil such as the 7-point Laplacian of BASELINE.json configs[3]):
                                     the values come from a shared list too and the slice streams nothing from the matrix arrays;
                                     bicg_constant_entries counts those entries. Same products, same order: bit-identical */ };
unsigned int bicg_ctx_flags(bicg_ctx *ctx);
/* bytes of MATRIX storage this context keeps on the GPU (CSR and/or sliced-ELL arrays, row pointers, offd block) */
unsigned long long bicg_device_matrix_bytes(bicg_ctx *ctx);
/* sliced-ELL entries (padding included) whose column indices the SpMV does not read (BICG_FLAG_UNIFORM), and the bytes one
 * SpMV streams from the matrix arrays (values + the column indices it does read + row pointers) */
unsigned long long bicg_uniform_entries(bicg_ctx *ctx);
unsigned long long bicg_constant_entries(bicg_ctx *ctx);
/* rows of MASKED slices: slices next to a grid face, whose rows are sub-sequences of one list of <= 16 (distance, value) pairs;
 * the product reads one 16-bit word per row for them (which pairs the row has) instead of values and columns. Counted in
 * bicg_uniform_entries / bicg_constant_entries too (with their padded entries). BICG_PLAN="masked=0" switches them off */
unsigned long long bicg_masked_rows(bicg_ctx *ctx);
/* The plane-marching product (csrc/bicg_stencil.hip): when the plan finds the 7-point stencil of a grid in the lists of a block
 * whose slices are all list-driven -- the interior's distances are (-sz, -sy, -1, 0, +1, +sy, +sz) with sy a multiple of 64 rows,
 * sz a multiple of sy, the rows a multiple of sz, and every other list a sub-sequence of that one -- y = A x runs with every
 * wavefront marching through the planes of its own grid lines (BASELINE.json configs[3]); same sums in the same order as
 * mult() (reference src/matrix.c:506-515). out = {in use (0/1), sy, sz / sy, rows / sz, lines per wavefront, planes per tile,
 * workgroups per product, x segments with masked slices}. BICG_PLAN="stencil=0" switches it off (slice-by-slice product);
 * BICG_PLAN="lines=2|4,planes=n" sets the tile; BICG_PLAN="ca-fuse=0" keeps CA-BiCGStab's q / y phase a kernel of its own */
int bicg_stencil_info(bicg_ctx *ctx, unsigned int out[8]);
/* Rows per lane of the plane-marching product: 1 (k_spmv_stencil), 2 or 4 (k_spmv_stencil_w: a wavefront's line is 128 / 256 rows,
 * one pair of edge lines per line instead of one per 64 rows; taken when every list of the block has the same value per distance --
 * constant-coefficient stencils -- and the vectors exceed the Infinity Cache, or BICG_PLAN="wide=2|4" asks for it; "wide=0" keeps
 * one row per lane). 0 when the product is not in use. Same sums as mult(), reference src/matrix.c:506-515. */
unsigned int bicg_stencil_rows_per_lane(bicg_ctx *ctx);
/* bicg_create_device_csr groups its list-driven slices by 64-bit hashes of their lists and then compares every slice with the
 * list it was given: the number of slices that did NOT match (hash collisions) and were put back on their stored columns and
 * values. 0 for contexts built by bicg_create (the host plan keys on the full lists). */
unsigned int bicg_plan_collisions(bicg_ctx *ctx);
/* Which product kernels this process has launched since the last call with reset != 0 (bit mask): 1 k_spmv_sell on padded slices,
 * 2 k_spmv_sell on jagged slices (columns gathered through the caches), 4 k_spmv_sell's loop over jagged slices with the x window
 * in LDS, 8 k_spmv_jagw (the three-trip form of that product, csrc/bicg_jagw.hip), 16 k_spmv_stencil, 32 k_spmv (CSR row blocks),
 * 64 k_spmv_rows (a row over several lanes), 128 a product with a pipelined phase in its epilogue,
 * 512 k_spmv_jagd (the three-trip product of jagged slices without a window: x gathered through the caches), 1024 k_spmv_jagw with a
 * list-driven window (the group's distinct columns one by one). Tests and bench.py assert on the kernel a matrix gets. */
unsigned int bicg_product_kernels(int reset);
/* 1 when the last bicg_solve_shifted / shifted_pipe_lopbicgstab call on this context ran its iterations as persistent launches
 * (k_shpipe_persist: latency-bound ranks, <= 32 shifts; BICG_PERSIST="shifted=0" keeps the multi-launch form) */
int bicg_last_shifted_persistent(bicg_ctx *ctx);
/* Which kernel the last bicg_spmm / bicg_shifted_residuals pass on this context ran: 2 the pipelined one (k_spmm_pipe,
 * csrc/bicg_spmm.hip: the x window of the next step copied global -> LDS by the DMA path while the current step multiplies,
 * persistent workgroups; padded 16-bit layouts whose distances fall into clusters), 1 the windowed one (k_spmm_win: the x values a
 * 256-row group touches staged in LDS through registers; BICG_PLAN="spmm-window=1" selects it everywhere, and layouts with x-window
 * runs take it), 0 the row-major kernel k_spmm_sell (BICG_PLAN="spmm-window=0", and layouts without clusters or window runs).
 * In all three the vectors' columns are bit-identical to bicg_spmv of each vector. */
int bicg_last_spmm_windowed(bicg_ctx *ctx);
unsigned long long bicg_spmv_matrix_bytes(bicg_ctx *ctx);

/* ---------------------------------------------------------------------------------------------
 * 5. Host-only helpers (no GPU needed; unit-tested on CPU).
 *   bicg_partition   <- reference src/matrix.c:295-308
 *   bicg_halo_plan   -- unique global columns referenced by an offd block, grouped by owner rank
 * ------------------------------------------------------------------------------------------- */
void bicg_partition(unsigned int n, int nranks, int *counts, int *displs);
/* Returns the halo length h and fills (caller-allocated, sizes noted):
 *   halo_cols[offd->nz upper bound]  ascending unique global columns
 *   recv_counts[nranks]              how many of them each rank owns
 *   renumbered[offd->nz]             offd->col mapped to local_rows + halo position */
int bicg_halo_plan(const CSR_Matrix *offd, const INFO_Matrix *info, int nranks, unsigned int local_rows,
                   unsigned int *halo_cols, int *recv_counts, unsigned int *renumbered);
/* Collective (through the caller's alltoallv): the SEND side of the halo exchange, in two steps.
 * Given this rank's halo_cols/recv_counts from bicg_halo_plan:
 *   bicg_halo_send_counts fills send_counts[nranks] (how many of OUR rows each rank needs) and
 *                         returns their sum;
 *   bicg_halo_send_lists  fills send_idx[sum] with the local row indices, grouped by destination
 *                         rank; returns the sum, or -1 if a peer asked for a row we do not own. */
int bicg_halo_send_counts(int nranks, const int *recv_counts, bicg_alltoallv_fn alltoallv, void *user,
                          int *send_counts);
int bicg_halo_send_lists(int rank, int nranks, const INFO_Matrix *info, unsigned int local_rows,
                         const unsigned int *halo_cols, const int *recv_counts, const int *send_counts,
                         bicg_alltoallv_fn alltoallv, void *user, unsigned int *send_idx);
/* greedy row blocks of at most chunk non-zeros (whole rows); returns the number of blocks and fills
 * rowblk[nblk+1] (caller provides rows+1 entries) */
unsigned int bicg_row_blocks(const unsigned int *ptr, unsigned int rows, unsigned int chunk,
                             unsigned int max_rows, unsigned int *rowblk);
/* x windows of the sliced-ELL plan for ragged rows: for every group of group_rows rows
 * (group_mask[g] != 0, or all when NULL) the columns its rows touch, merged into runs of consecutive columns
 * (gaps of <= gap unused columns are bridged). runs[2i] = first column, runs[2i+1] = (first slot << 16) | length;
 * win_ptr[g] .. win_ptr[g+1] index a group's runs; *slots_used = slots of the largest window. Returns the
 * number of runs, -1 when a group needs more than max_slots slots. runs / win_ptr / slots_used may be NULL
 * (count only). bicg_window_slot: the slot of column c in the window runs[2*first .. 2*end). */
long bicg_window_plan(const unsigned int *ptr, const unsigned int *col, unsigned int rows, unsigned int group_rows,
                      const char *group_mask, unsigned int max_slots, unsigned int gap, unsigned int *win_ptr,
                      unsigned int *runs, unsigned int *slots_used);
unsigned int bicg_window_slot(const unsigned int *runs, unsigned int first, unsigned int end, unsigned int c);
/* Host threads of the set-up (bicg_create's plan and bicg_window_plan cut their loops over slices / groups / rows into one range
 * per thread; the results do not depend on the number). 0 < n: use n threads from now on; n < 0: back to the default; returns the
 * number in use. Default: BICG_PLAN_THREADS, else the hardware threads this process may run on (its affinity mask) divided by the
 * ranks that share the host (LOCAL_WORLD_SIZE / OMPI_COMM_WORLD_LOCAL_SIZE / MPI_LOCALNRANKS / SLURM_NTASKS_PER_NODE, else the
 * communicator's size), at most 32. */
int bicg_set_plan_threads(int n);

/* Plan of the persistent iteration (host only, what bicg_create builds for latency-bound ranks): the
 * workgroups' slices as padded entries {value, 16-bit slot of the column in the workgroup's window}, diag entries first, then
 * the offd entries (offd_renumbered: columns = local rows + halo position as bicg_halo_plan renumbers them; NULL for one rank),
 * and the window runs {first column, (first slot << 16) | length} per workgroup. gmax = workgroups available (CUs - 1).
 * summary = {slices per workgroup, workgroups, window slots, most runs of a workgroup, most entries of a workgroup, entries,
 * runs, 0}; arrays may be NULL (sizes: a first call). Returns 0 when the block qualifies (<= 15 slices per workgroup, windows
 * within 16 384 slots). */
int bicg_persist_plan(const CSR_Matrix *diag, const CSR_Matrix *offd_renumbered, unsigned int gmax, unsigned int summary[8],
                      unsigned int *pbase, unsigned short *pslot, double *pval, unsigned short *rlen, unsigned short *rdiag,
                      unsigned int *win_ptr, unsigned int *win_runs);

/* Matrix-Market block loader (host only): what MPI_csr_load_matrix_block produces for `rank` of
 * `nranks` (reference src/matrix.c:402-419) -- diag block with local columns, offd block with global
 * columns, file order inside a row, equal-rows partition -- reading the file once. Arrays are
 * malloc'ed; release with bicg_mtx_free. Returns 0 on success. (The C host additionally has an MPI
 * variant in which every rank tokenises 1/P of the file, mpi-bicgstab_amd/host/bicg_mtx.h.) */
int bicg_mtx_load_block(const char *path, int rank, int nranks, CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info);
/* The "next" pieces of the ingest path, host only:
 *   bicg_partition_nnz        contiguous row blocks with equal non-zero counts (idea of the reference's
 *                             abandoned DYNAMIC_ROWS branch, archive/matrix.c:407-446); the solver accepts
 *                             any contiguous partition through INFO_Matrix.recvcounts/displs
 *   bicg_mtx_load_block_part  the loader above with part = BICG_PART_ROWS | BICG_PART_NNZ
 *   bicg_mtx_cache_save/load  checksummed binary copy of one rank's parsed blocks; load returns 0 on a
 *                             valid hit for exactly this (rank, nranks, part) and unchanged source file */
/*   bicg_mtx_parse_double    the loader's conversion of one value: returns 0 when its own exact fast path (Eisel-Lemire,
 *                             <= 19 significant digits) produced it, 1 when it was handed to strtod; either way *value is
 *                             what fscanf("%lg") gives (src/matrix.c:333) and *consumed the characters used */
enum { BICG_PART_ROWS = 0, BICG_PART_NNZ = 1 };
void bicg_partition_nnz(const unsigned int *row_nnz, unsigned int n, int nranks, int *counts, int *displs);
int bicg_mtx_parse_double(const char *text, double *value, int *consumed);
int bicg_mtx_load_block_part(const char *path, int rank, int nranks, int part, CSR_Matrix *diag, CSR_Matrix *offd,
                             INFO_Matrix *info);
int bicg_mtx_cache_save(const char *cache_path, const char *src_path, int rank, int nranks, int part,
                        const CSR_Matrix *diag, const CSR_Matrix *offd, const INFO_Matrix *info);
int bicg_mtx_cache_load(const char *cache_path, const char *src_path, int rank, int nranks, int part,
                        CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info);
void bicg_mtx_free(CSR_Matrix *diag, CSR_Matrix *offd, INFO_Matrix *info);
/* Device-side COO -> CSR (needs a GPU): the triplets this rank owns -- FILE order, global row and column
 * indices, rows in [lo, hi) -- become the diag block (local columns) and the offd block (global columns,
 * cols = ncols) that MPI_csr_load_matrix_block yields (reference src/matrix.c:336-392), file order kept inside
 * every row (a stable sort on the row key, like the reference's merge sort, src/matrix.c:135-183).
 * Arrays are malloc'ed; release with free() / bicg_mtx_free. bicg_mtx_set_block_builder(fn) makes the
 * Matrix-Market loaders above use fn for that step (NULL = host counting sort); the C host does so with
 * BICG_INGEST=device. */
typedef int (*bicg_block_builder_fn)(const unsigned int *row, const unsigned int *col, const double *val, unsigned long nnz,
                                     unsigned int lo, unsigned int hi, unsigned int ncols, CSR_Matrix *diag, CSR_Matrix *offd);
int bicg_coo_to_blocks_device(const unsigned int *row, const unsigned int *col, const double *val, unsigned long nnz,
                              unsigned int lo, unsigned int hi, unsigned int ncols, CSR_Matrix *diag, CSR_Matrix *offd);
void bicg_mtx_set_block_builder(bicg_block_builder_fn fn);

const char *bicg_version(void);
/* 1: the library was built with `make EXPERIMENTS=1` and reads the measurement knobs of the development rounds (csrc/bicg_knobs.h;
 * the negative results they select -- window-fused plain iteration, direct SpMM -- are compiled in); 0: the default build */
int bicg_has_experiments(void);
/* What the library reads from one of its token-list variables (BICG_PLAN, BICG_PERSIST, BICG_TEST: comma-separated `name` or
 * `name=value` tokens): copies the value of token `name` in $set to out ("1" for a bare token) and returns
 * its length, -1 when the variable or the token is absent. No device needed. */
int bicg_switch_value(const char *set, const char *name, char *out, int cap);
/* Tokens of $set that are none of the names the library knows for it (a typing error selects nothing): copies the first one to out
 * and returns their number. bicg_create prints one line per list on rank 0 when there are any. */
int bicg_switch_unknown(const char *set, char *out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* BICGSTAB_HIP_H */
