/* r_emul.c -- TEST INFRASTRUCTURE: a small EXECUTABLE emulation of the part of R's C API that r/harmony_mi355x_glue.c uses
 * (declared in tests/stubs/R.h).  R is not installed in this image, so the glue cannot be built into a package; linked against this
 * file it can at least RUN: tests/test_abi_cpu.py drives its error path on the CPU, tests/test_gpu_parity2.py the whole
 * setup -> init -> cluster / correct loop -> getters sequence of r/harmony_mi355x.R on the GPU, through the same .Call entry points
 * with the same argument marshalling, and compares with the ctypes path bit for bit.
 *
 * What is emulated: SEXPs as heap records (REALSXP / INTSXP / LGLSXP / STRSXP / CHARSXP / EXTPTRSXP / NILSXP with a `dim` attribute),
 * PROTECT as a no-op (nothing is ever collected: the test process is short-lived), Rf_error as a longjmp back to emul_call() with the
 * formatted message kept, Rf_warning as a message list, finalizers run by emul_release(), R_CheckUserInterrupt as a flag the test sets,
 * GetRNGstate / PutRNGstate / unif_rand as R's default generator (MT19937 after set.seed's scrambling; emul_set_seed = set.seed).
 * What is NOT: everything else of R.  This shows that the glue's marshalling and control flow work -- not that it links against libR. */
#define _POSIX_C_SOURCE 200809L
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "R.h"
#include "R_ext/Rdynload.h"

#define NILSXP 0
#define CHARSXP 9
#define LGLSXP 10
#define STRSXP 16
#define EXTPTRSXP 22

struct SEXPREC {
  int type;
  R_xlen_t length;
  void* data;              /* REAL / INTEGER / LOGICAL payload, char* of a CHARSXP, SEXP* of a STRSXP, the address of an external pointer */
  SEXP dim;                /* INTSXP of length 2 or NULL */
  R_CFinalizer_t finalizer;
};
static struct SEXPREC nil_rec = {NILSXP, 0, NULL, NULL, NULL};
static struct SEXPREC dimsym_rec = {NILSXP, 0, NULL, NULL, NULL};
SEXP R_NilValue = &nil_rec;
SEXP R_DimSymbol = &dimsym_rec;

static jmp_buf* active_jmp = NULL;
static char last_error[1024];
static char warnings[4096];
static int n_warnings = 0, interrupt_pending = 0, rng_state_held = 0, rng_brackets = 0;

static SEXP mk(int type, R_xlen_t n, size_t elem) {
  SEXP s = (SEXP)calloc(1, sizeof(struct SEXPREC));
  s->type = type; s->length = n; s->data = n ? calloc((size_t)n, elem) : NULL;
  return s;
}
double* REAL(SEXP s) { if (s->type != REALSXP) Rf_error("REAL() can only be applied to a 'numeric', not a type %d", s->type); return (double*)s->data; }
int* INTEGER(SEXP s) { if (s->type != INTSXP && s->type != LGLSXP) Rf_error("INTEGER() can only be applied to a 'integer', not a type %d", s->type); return (int*)s->data; }
int LENGTH(SEXP s) { return (int)s->length; }
R_xlen_t XLENGTH(SEXP s) { return s->length; }
int TYPEOF(SEXP s) { return s->type; }
SEXP STRING_ELT(SEXP s, R_xlen_t i) { if (s->type != STRSXP || i >= s->length) Rf_error("STRING_ELT() on a non-character vector or out of range"); return ((SEXP*)s->data)[i]; }
const char* CHAR(SEXP s) { if (s->type != CHARSXP) Rf_error("CHAR() can only be applied to a 'CHARSXP'"); return (const char*)s->data; }
SEXP PROTECT(SEXP s) { return s; }
void UNPROTECT(int n) { (void)n; }
double Rf_asReal(SEXP s) {
  if (s->length < 1) return 0.0 / 0.0;
  if (s->type == REALSXP) return ((double*)s->data)[0];
  if (s->type == INTSXP || s->type == LGLSXP) return (double)((int*)s->data)[0];
  Rf_error("asReal: unsupported type %d", s->type); return 0;
}
int Rf_asInteger(SEXP s) { return (int)Rf_asReal(s); }
int Rf_asLogical(SEXP s) { return Rf_asReal(s) != 0; }
SEXP Rf_allocVector(unsigned int type, R_xlen_t n) {
  if (type == REALSXP) return mk(REALSXP, n, sizeof(double));
  if (type == INTSXP || type == LGLSXP) return mk((int)type, n, sizeof(int));
  Rf_error("allocVector: unsupported type %u", type); return R_NilValue;
}
SEXP Rf_allocMatrix(unsigned int type, int nr, int nc) {
  SEXP s = Rf_allocVector(type, (R_xlen_t)nr * nc);
  s->dim = mk(INTSXP, 2, sizeof(int)); ((int*)s->dim->data)[0] = nr; ((int*)s->dim->data)[1] = nc;
  return s;
}
SEXP Rf_ScalarLogical(int v) { SEXP s = mk(LGLSXP, 1, sizeof(int)); ((int*)s->data)[0] = v != 0; return s; }
SEXP Rf_ScalarInteger(int v) { SEXP s = mk(INTSXP, 1, sizeof(int)); ((int*)s->data)[0] = v; return s; }
SEXP Rf_ScalarReal(double v) { SEXP s = mk(REALSXP, 1, sizeof(double)); ((double*)s->data)[0] = v; return s; }
SEXP Rf_getAttrib(SEXP s, SEXP name) { if (name == R_DimSymbol && s->dim) return s->dim; return R_NilValue; }
int Rf_isNull(SEXP s) { return s == R_NilValue || s->type == NILSXP; }
void Rf_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(last_error, sizeof(last_error), fmt, ap); va_end(ap);
  if (active_jmp) longjmp(*active_jmp, 1);
  fprintf(stderr, "Rf_error outside emul_call: %s\n", last_error); abort();
}
void Rf_warning(const char* fmt, ...) {
  char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  n_warnings++;
  if (strlen(warnings) + strlen(buf) + 2 < sizeof(warnings)) { strcat(warnings, buf); strcat(warnings, "\n"); }
}
void* R_ExternalPtrAddr(SEXP s) { return s->type == EXTPTRSXP ? s->data : NULL; }
SEXP R_MakeExternalPtr(void* p, SEXP tag, SEXP prot) { (void)tag; (void)prot; SEXP s = mk(EXTPTRSXP, 0, 1); s->data = p; return s; }
void R_ClearExternalPtr(SEXP s) { s->data = NULL; }
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t f, Rboolean onexit) { (void)onexit; s->finalizer = f; }
void R_CheckUserInterrupt(void) { if (interrupt_pending) { interrupt_pending = 0; Rf_error("interrupted"); } }
/* runs fun(data) in a context of its own: FALSE when it left through an error (here: the pending interrupt) */
Rboolean R_ToplevelExec(void (*fun)(void*), void* data) {
  jmp_buf here; jmp_buf* outer = active_jmp; Rboolean ok = TRUE;
  active_jmp = &here;
  if (setjmp(here) == 0) fun(data); else ok = FALSE;
  active_jmp = outer;
  return ok;
}

/* ---- R's default generator (Mersenne Twister after set.seed's LCG scrambling), written for this file -------------------------------- */
static uint32_t mt[624]; static int mti = 625;
void emul_set_seed(uint32_t seed) {
  for (int j = 0; j < 50; j++) seed = 69069u * seed + 1u;
  seed = 69069u * seed + 1u;
  for (int j = 0; j < 624; j++) { seed = 69069u * seed + 1u; mt[j] = seed; }
  mti = 624;
}
static uint32_t genrand(void) {
  if (mti >= 624) {
    if (mti == 625) emul_set_seed(4357u);
    for (int k = 0; k < 624; k++) { uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu); mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
    mti = 0;
  }
  uint32_t y = mt[mti++];
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}
void GetRNGstate(void) { rng_state_held++; rng_brackets++; }
void PutRNGstate(void) { rng_state_held--; }
double unif_rand(void) {
  if (rng_state_held <= 0) Rf_error("unif_rand() outside GetRNGstate() / PutRNGstate()");
  const double x = (double)genrand() * 2.3283064365386963e-10, h = 0.5 * 2.328306437080797e-10;
  return x <= 0.0 ? h : ((1.0 - x) <= 0.0 ? 1.0 - h : x);
}

/* ---- what the test (the stand-in for the R interpreter) uses ------------------------------------------------------------------------ */
SEXP emul_real(const double* v, R_xlen_t n, int nrow, int ncol) {     /* numeric vector (nrow = 0) or matrix */
  SEXP s = nrow ? Rf_allocMatrix(REALSXP, nrow, ncol) : Rf_allocVector(REALSXP, n);
  if (n) memcpy(s->data, v, sizeof(double) * (size_t)n);
  return s;
}
SEXP emul_int(const int* v, R_xlen_t n, int nrow, int ncol) {
  SEXP s = nrow ? Rf_allocMatrix(INTSXP, nrow, ncol) : Rf_allocVector(INTSXP, n);
  if (n) memcpy(s->data, v, sizeof(int) * (size_t)n);
  return s;
}
SEXP emul_logical(int v) { return Rf_ScalarLogical(v); }
SEXP emul_string(const char* c) {
  SEXP ch = mk(CHARSXP, (R_xlen_t)strlen(c), 1); free(ch->data); ch->data = strdup(c);
  SEXP s = mk(STRSXP, 1, sizeof(SEXP)); ((SEXP*)s->data)[0] = ch;
  return s;
}
int emul_type(SEXP s) { return s->type; }
R_xlen_t emul_length(SEXP s) { return s->length; }
const void* emul_data(SEXP s) { return s->data; }
int emul_dim(SEXP s, int which) { return s->dim ? ((int*)s->dim->data)[which] : -1; }
const char* emul_last_error(void) { return last_error; }
const char* emul_warnings(void) { return warnings; }
int emul_warning_count(void) { return n_warnings; }
int emul_rng_brackets(void) { return rng_brackets; }
void emul_clear(void) { last_error[0] = 0; warnings[0] = 0; n_warnings = 0; rng_brackets = 0; }
void emul_set_interrupt(int on) { interrupt_pending = on; }
void emul_release(SEXP s) { if (s && s->type == EXTPTRSXP && s->finalizer) s->finalizer(s); }       /* what the garbage collector would do */
/* .Call(fn, args...): NULL when the callee left through Rf_error (message: emul_last_error) */
typedef SEXP (*F0)(void); typedef SEXP (*F1)(SEXP); typedef SEXP (*F2)(SEXP, SEXP); typedef SEXP (*F3)(SEXP, SEXP, SEXP); typedef SEXP (*F4)(SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*F18)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP emul_call(void* fn, int nargs, SEXP* a) {
  jmp_buf here; SEXP out = NULL;
  active_jmp = &here;
  if (setjmp(here) == 0) {
    switch (nargs) {
      case 0: out = ((F0)fn)(); break;
      case 1: out = ((F1)fn)(a[0]); break;
      case 2: out = ((F2)fn)(a[0], a[1]); break;
      case 3: out = ((F3)fn)(a[0], a[1], a[2]); break;
      case 4: out = ((F4)fn)(a[0], a[1], a[2], a[3]); break;
      case 18: out = ((F18)fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17]); break;
      default: snprintf(last_error, sizeof(last_error), "emul_call: %d arguments not supported", nargs); out = NULL;
    }
  } else out = NULL;
  active_jmp = NULL;
  return out;
}
/* R_registerRoutines / R_useDynamicSymbols: the registration table is looked up by name, as .Call("name", ...) does */
static const R_CallMethodDef* registered = NULL;
int R_registerRoutines(DllInfo* dll, const void* c, const R_CallMethodDef* call, const void* f, const void* e) { (void)dll; (void)c; (void)f; (void)e; registered = call; return 1; }
Rboolean R_useDynamicSymbols(DllInfo* dll, Rboolean v) { (void)dll; (void)v; return TRUE; }
void* emul_lookup(const char* name, int* nargs) {
  for (const R_CallMethodDef* m = registered; m && m->name; m++) if (!strcmp(m->name, name)) { *nargs = m->numArgs; return (void*)m->fun; }
  return NULL;
}
