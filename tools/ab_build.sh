#!/bin/bash
# tools/ab_build.sh <name> [-D flags...]: build the library with extra flags into qcat_amd/csrc/build/ab/<name>.so
# (the default library is rebuilt last by the caller: python __graft_entry__.py)
cd $(dirname $0)/..
name=$1; shift
mkdir -p qcat_amd/csrc/build/ab
QCAT_EXTRA_HIPFLAGS="$*" python -c "import __graft_entry__ as g; g.build(force=True)" 2>&1 | grep -iE "error" | head -5
cp qcat_amd/csrc/libqcat_hip.so qcat_amd/csrc/build/ab/$name.so && echo "built $name"
