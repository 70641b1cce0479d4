#!/bin/bash
# Assembles the R package `harmonymi355x` (the recommended binding) from this repository's single sources:
#     r/make_companion_package.sh [output directory, default ./harmonymi355x]  &&  R CMD INSTALL harmonymi355x
# and prints the ONE line of the reference package that changes (R/ui.R:269).  Needs no R to assemble; R + a C compiler to install.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(dirname "$HERE"); OUT=${1:-harmonymi355x}
mkdir -p "$OUT/R" "$OUT/src"
cp "$HERE/companion/DESCRIPTION" "$HERE/companion/NAMESPACE" "$OUT/"
cp "$HERE/harmony_mi355x.R" "$OUT/R/"
cp "$HERE/harmony_mi355x_glue.c" "$OUT/src/"
sed "s|@HMX_ROOT@|$ROOT|" "$HERE/companion/Makevars" > "$OUT/src/Makevars"
cat <<EOF
assembled $OUT (R CMD INSTALL $OUT).  In the reference package change ONE line, R/ui.R:269:
    harmonyObj <- new(harmony)
 -> harmonyObj <- if (requireNamespace("harmonymi355x", quietly = TRUE) && isTRUE(getOption("harmony.mi355x", TRUE))) harmonymi355x::new_harmony_mi355x() else new(harmony)
(r/ui_R_269.sed does it: sed -i -f r/ui_R_269.sed <harmony>/R/ui.R).  Nothing else in the reference package changes; options(harmony.mi355x = FALSE) gives its own CPU engine back.
EOF
