#!/bin/bash
# build_ref.sh -- builds the REAL reference (from its own sources, where they lie under $REF) into oracle/_ref/ when,
# and only when, everything it includes is present.  No stand-ins: the reference's translation units include Boost
# (data.cpp:17, utilities.cpp:8, vamp.cpp:19: <boost/math/distributions/students_t.hpp>; options.cpp:9:
# <boost/algorithm/string/trim.hpp>; vamp_probit.cpp:14-17: <boost/numeric/ublas/...>) and <mpi.h>.  Without real Boost
# headers this script says so and builds nothing: the oracle then stays "parity unpinned" (docs/history/rounds1-3.md section 2).
#
#   REF=/root/reference BOOST_ROOT=/path/to/boost oracle/ref_recipe/build_ref.sh
#
# exit 0 = built (binaries in oracle/_ref/), exit 3 = reference unbuildable here, anything else = a build error.
set -u
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../_ref"
REF="${REF:-/root/reference}"
CXX="${CXX:-g++}"

if [ ! -f "$REF/vamp.cpp" ]; then
  echo "reference tree absent ($REF): nothing to build; keeping whatever oracle/_ref holds"
  exit 3
fi

# ---- Boost: real headers only
BOOST_INC=""
for cand in "${BOOST_ROOT:-}" "${BOOST_ROOT:-}/include" /usr/include /usr/local/include /opt/conda/include /opt/boost/include; do
  [ -n "$cand" ] || continue
  if [ -f "$cand/boost/math/distributions/students_t.hpp" ] && [ -f "$cand/boost/numeric/ublas/lu.hpp" ] && \
     [ -f "$cand/boost/algorithm/string/trim.hpp" ] && [ -f "$cand/boost/version.hpp" ]; then
    # a genuine Boost tree carries its version macro; a directory of hand-written stubs does not
    if grep -q "define BOOST_LIB_VERSION" "$cand/boost/version.hpp"; then BOOST_INC="$cand"; break; fi
  fi
done
if [ -z "$BOOST_INC" ]; then
  echo "reference unbuildable: parity unpinned"
  echo "  (no Boost headers found in \$BOOST_ROOT, /usr/include, /usr/local/include, /opt/conda/include; the reference's"
  echo "   data.cpp / vamp.cpp / utilities.cpp / options.cpp / vamp_probit.cpp include them, and stand-ins are not allowed)"
  exit 3
fi

# ---- MPI: headers + library (the conda MPICH of this image; its mpic++ wrapper points at a missing cross-compiler, so g++
# is called directly and libmpi is linked through a private directory to keep conda's old libstdc++ out of the link)
MPI_INC=""; MPI_LIBDIR=""
for cand in "${MPI_ROOT:-}" /opt/conda /usr /usr/lib/x86_64-linux-gnu/openmpi /usr/lib/x86_64-linux-gnu/mpich; do
  [ -n "$cand" ] || continue
  if [ -f "$cand/include/mpi.h" ]; then MPI_INC="$cand/include"; fi
  for l in "$cand/lib" "$cand/lib64" "$cand/lib/x86_64-linux-gnu"; do
    if ls "$l"/libmpi.so* >/dev/null 2>&1; then MPI_LIBDIR="$l"; fi
  done
  [ -n "$MPI_INC" ] && [ -n "$MPI_LIBDIR" ] && break
done
if [ -z "$MPI_INC" ] || [ -z "$MPI_LIBDIR" ]; then
  echo "reference unbuildable: parity unpinned  (no MPI headers / library found; set MPI_ROOT)"
  exit 3
fi
mkdir -p "$OUT/mpilib"
for f in "$MPI_LIBDIR"/libmpi.so* "$MPI_LIBDIR"/libgfortran.so* "$MPI_LIBDIR"/libquadmath.so*; do
  [ -e "$f" ] && ln -sf "$f" "$OUT/mpilib/"
done

# README.md:24's one-line recipe, minus the site wrapper.  -include cstring / iomanip: vamp.cpp uses strcmp and
# std::setprecision (:173, :1191) and relies on Boost's headers to pull them in on some versions.
FLAGS="-std=c++17 -O2 -march=native -fopenmp -include cstring -include iomanip -I$BOOST_INC -I$MPI_INC -I$REF"
LINK="-L$OUT/mpilib -lmpi -Wl,-rpath,$OUT/mpilib -lstdc++fs"
CORE="$REF/vamp.cpp $REF/utilities.cpp $REF/data.cpp $REF/options.cpp"
set -e
build() {  # name extra-flags sources...
  local name=$1 extra=$2; shift 2
  echo "[ref] $name"
  $CXX $FLAGS $extra "$@" -o "$OUT/$name" $LINK
}
# vamp.cpp #includes vamp_probit.cpp / vamp_Huber.cpp / denoiserXXT.cpp itself (vamp.cpp:15-17): the README's five files
ALL="$CORE"
build sim_scalar        ""          $REF/sim.cpp $ALL               # scalar path: mask / pad / guard semantics
build sim_manvect       "-DMANVECT" $REF/sim.cpp $ALL               # AVX-512 path, for the cross-check of 8c
build main_real         ""          $REF/main_real.cpp $ALL
build main_real_probit  ""          $REF/main_real_probit.cpp $ALL
build sim_probit        ""          $REF/sim_probit.cpp $ALL
build harness           ""          "$HERE/harness.cpp" $ALL
echo "[ref] built into $OUT (Boost: $BOOST_INC, MPI: $MPI_INC)"
date -u +%FT%TZ > "$OUT/BUILT"
exit 0
