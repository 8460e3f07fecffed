#!/bin/bash
# usage: scripts/gpu_r2.sh <tag> <cmd...> : runs a shell command on the box with gpurun_out/<tag>/ created, output to log
tag=$1; shift
mkdir -p gpurun_out/$tag
export OUT=gpurun_out/$tag
bash -c "$*"
