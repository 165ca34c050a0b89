# r/zzz.R -- loads the MI355X engine when the package is loaded. Goes into the reference as R/zzz.R next to
# `useDynLib(ppcseq, .registration = TRUE)` (NAMESPACE:143), whose Rcpp module `stan_fit4negBinomial_MPI_mod`
# (src/RcppExports.cpp:15-25, R/stanmodels.R:7-25) it makes unnecessary for the inference path.
# libppcx.so exports a plain C ABI (include/ppcx.h): no Rcpp glue, no registration -- .C() finds `ppcx_do_inference_C`
# by name. R is not installed in the build environment of this repository: this file is source, not exercised here; the same
# call with the same argument shapes is exercised by the plain-C host tests/c_host/dot_c_host.c.
.onLoad <- function(libname, pkgname) {
  so <- Sys.getenv("PPCX_LIB", file.path(libname, pkgname, "libs", "libppcx.so"))
  dyn.load(so)                                   # exports ppcx_* (C ABI)
}

.onUnload <- function(libpath) {
  so <- Sys.getenv("PPCX_LIB", file.path(libpath, "libs", "libppcx.so"))
  if (so %in% vapply(getLoadedDLLs(), function(d) d[["path"]], "")) dyn.unload(so)
}
