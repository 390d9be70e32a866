/* oracle/_ref build shim (test infrastructure, NOT product, NOT a copy of srsRAN).
 * /root/reference/lib/include/falcon/util/rnti_manager_c.h:30 includes "srsran/phy/phch/dci.h" and uses nothing from it; srsRAN itself is an
 * un-vendored dependency that is absent here.  This empty header lets the reference's own RNTIManager.cc / Histogram.cc / Interval.cc compile
 * from where they lie (oracle/Makefile.ref).  It declares nothing on purpose: a reference file that needs a real srsRAN declaration must fail
 * to compile against it, and is then "unbuildable" (DESIGN.md section 2). */
#pragma once
