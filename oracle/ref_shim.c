/*
 * ref_shim.c -- ORACLE build helper (test infrastructure, NOT product code).
 *
 * Linked into the oracle/_ref builds of the real reference with -Wl,--wrap=malloc so that every
 * malloc() made by the reference objects returns zeroed memory. The reference's ca_/pipe_ solvers
 * read p, s, z, v before writing them (reference src/solver.c:217-222, 352-360; SURVEY.md
 * section 4 defect 1) and only work when malloc happens to hand out fresh zero pages; this makes
 * that assumption deterministic for small test matrices without touching the reference sources.
 * (The uninitialised scalar `omega` is handled by compiling with -ftrivial-auto-var-init=zero.)
 */
#include <stdlib.h>

void *__wrap_malloc(size_t bytes) { return calloc(1, bytes ? bytes : 1); }
