# the one line of the reference package that changes (R/ui.R:269): ask the companion package for the engine object when it is installed
s|^\( *\)harmonyObj <- new(harmony) *$|\1harmonyObj <- if (requireNamespace("harmonymi355x", quietly = TRUE) \&\& isTRUE(getOption("harmony.mi355x", TRUE))) harmonymi355x::new_harmony_mi355x() else new(harmony)|
