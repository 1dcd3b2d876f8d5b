#!/bin/bash
# two processes on GPU 0: does a kernel of A see what a kernel of B writes through an IPC mapping, while A spins?
D=$(mktemp -d)
HERE=$(dirname $0)
[ -x $HERE/ipc_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $HERE/ipc_probe $HERE/ipc_probe.hip || exit 1
timeout -k 5 30 $HERE/ipc_probe A $D 0 > $D/a.log 2>&1 &
PA=$!
timeout -k 5 30 $HERE/ipc_probe B $D 0 > $D/b.log 2>&1
RB=$?
wait $PA
RA=$?
cat $D/a.log $D/b.log
echo "exit A=$RA B=$RB"
rm -rf $D
