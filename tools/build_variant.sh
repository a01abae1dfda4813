# usage: bash tools/build_variant.sh <name> <extra hipcc flags...>   -> ab/libsynthray_<name>.so (trace.hip rebuilt with the flags)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); name=$1; shift
make -s -j8 -C $R/synthpy_amd/csrc
mkdir -p $R/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -I$R/include -Wall -Wno-unused-function "$@" -c $R/synthpy_amd/csrc/trace.hip -o /tmp/trace_$name.o
cd $R/synthpy_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $R/ab/libsynthray_$name.so runtime.o volume.o /tmp/trace_$name.o deposit.o comm.o field.o beam.o -ldl
echo built ab/libsynthray_$name.so
