#!/bin/bash
# A/B build of the engine as of a git ref -> scripts/_bin/libfe_<name>.so.  usage: scripts/build_ref.sh <name> <git-ref> [-D...]
NAME=$1; REF=$2; shift 2
D=/tmp/fe_ref_$NAME; mkdir -p $D/fluidlab_amd/csrc $D/include
cd "$(dirname "$0")/.." && for f in fluidlab_amd/csrc/fe_engine.hip fluidlab_amd/csrc/fe_math.h fluidlab_amd/csrc/fe_smoke.h fluidlab_amd/csrc/fe_mesh.h include/fluidengine.h; do git show $REF:$f > $D/$f; done
SRC=$D scripts/build_variant.sh $NAME "$@"
