#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 300 tools/bin/xpose_proto 128 > $O/xpose_proto_128.txt 2>&1; cat $O/xpose_proto_128.txt
timeout 300 tools/bin/xpose_proto 96 > $O/xpose_proto_96.txt 2>&1; cat $O/xpose_proto_96.txt
