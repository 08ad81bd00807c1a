#!/bin/bash
# Ablation builds of the product library for kernel experiments (never shipped): build/abl/<name>.so
# usage: tools/ablate.sh name "-DFLAG1 -DFLAG2"
set -e
cd "$(dirname "$0")/.."
mkdir -p build/abl
make -C tempestmodel_amd/csrc -j8 EXTRA="$2" BUILD="$PWD/build/abl/obj_$1" OUT="$PWD/build/abl/$1.so" > /dev/null
echo "built build/abl/$1.so"
