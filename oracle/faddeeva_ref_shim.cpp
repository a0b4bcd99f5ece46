// C shim over the reference's vendored Faddeeva package (cpp_source/Faddeeva.cc, compiled in
// place from /root/reference by oracle/Makefile into oracle/_ref/). Test infrastructure only.
#include "Faddeeva/Faddeeva.hh"
extern "C" {
double ref_erfcx(double x) { return Faddeeva::erfcx(x); }
double ref_erf(double x) { return Faddeeva::erf(x); }
double ref_erfc(double x) { return Faddeeva::erfc(x); }
}
