/* Minimal stand-in for R's C API -- ONLY for the syntax/type check of r/harmony_mi355x_glue.c in tests/test_abi_cpu.py
 * (R is not installed in this image).  Declarations follow Rinternals.h / R_ext/Rdynload.h of R 4.x. */
#ifndef STUB_R_H
#define STUB_R_H
#include <stddef.h>
typedef struct SEXPREC* SEXP;
typedef int Rboolean;
typedef ptrdiff_t R_xlen_t;
#define TRUE 1
#define FALSE 0
extern SEXP R_NilValue, R_DimSymbol;
#define REALSXP 14
#define INTSXP 13
double* REAL(SEXP);
int* INTEGER(SEXP);
int LENGTH(SEXP);
R_xlen_t XLENGTH(SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
const char* CHAR(SEXP);
SEXP PROTECT(SEXP);
void UNPROTECT(int);
double Rf_asReal(SEXP);
int Rf_asInteger(SEXP);
int Rf_asLogical(SEXP);
SEXP Rf_allocVector(unsigned int, R_xlen_t);
SEXP Rf_allocMatrix(unsigned int, int, int);
int TYPEOF(SEXP);
SEXP Rf_ScalarLogical(int);
SEXP Rf_ScalarInteger(int);
SEXP Rf_ScalarReal(double);
SEXP Rf_getAttrib(SEXP, SEXP);
int Rf_isNull(SEXP);
void Rf_error(const char*, ...);
void Rf_warning(const char*, ...);
void* R_ExternalPtrAddr(SEXP);
SEXP R_MakeExternalPtr(void*, SEXP, SEXP);
void R_ClearExternalPtr(SEXP);
typedef void (*R_CFinalizer_t)(SEXP);
void R_RegisterCFinalizerEx(SEXP, R_CFinalizer_t, Rboolean);
Rboolean R_ToplevelExec(void (*fun)(void*), void* data);
void R_CheckUserInterrupt(void);
/* R_ext/Random.h */
void GetRNGstate(void);
void PutRNGstate(void);
double unif_rand(void);
#endif
