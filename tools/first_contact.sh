#!/bin/bash
# First contact with a multi-GPU node: the exact sequence to run when an 8-GPU MI355X node is available, each line with
# what it must print.  No message of this engine has crossed two devices yet (the development lease is one GPU); every
# step below is the smallest run that exercises one more piece, so that a failure names its cause.
#   usage: tools/first_contact.sh [N_MAX]       (from the repo root; N_MAX = 8)
# Every bench line at N > 1 carries its own parity proof ("multi_rank": {"parity": {"identical_to_single_rank_engine": true}}):
# after the timing all ranks step three times from the initial state, rank 0 also steps a single-rank engine holding all 24
# patches, and rank 0's patches must be identical, bit for bit.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
NMAX=${1:-8}
PORT=${PORT:-29541}
run() {   # ranks, extra bench flags...
	local n=$1; shift
	python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $n "$@"
}
field() {   # prints the expression of the run's JSON line, or says that the run left none (failed / refused) instead of a traceback
	python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls:
    print('   (no JSON line: the run failed or was refused -- see its .err file)')
else:
    d = json.loads(ls[-1]); print($1)"
}

echo "== 0. device count (expect >= 2)"
python3 -c "import torch; print(torch.cuda.device_count())"

echo "== 0b. the driver's spelling: python bench.py --gpus 2 WITHOUT a launcher (bench.py starts torch.distributed.run itself, as a child process)"
echo "   expect: stderr says \"without a launcher\", ONE JSON line with n_gpus 2 and multi_rank.per_rank_kernel_ms_per_step of length 2"
python bench.py --gpus 2 --steps 3 --warmup 1 --develop 1 --sustained-steps 0 --no-cpu-baseline 2>gpurun_out/first_contact_0b.err | tee gpurun_out/first_contact_0b.json | \
	field "(d['n_gpus'], len(d['multi_rank']['per_rank_kernel_ms_per_step']), d['multi_rank']['parity'])"

echo "== 1. two ranks, RCCL send/recv, 3 steps, NCCL_DEBUG=INFO"
echo "   expect: 'NCCL INFO ... comm ... nranks 2' from both ranks, then ONE JSON line with"
echo "           multi_rank.rccl_comm_ranks == 2, multi_rank.halo_transport == 'rccl', multi_rank.parity.identical_to_single_rank_engine == true,"
echo "           rank0_message_bytes_per_peer: one peer, (5 L + 1) slabs x 8 B x boundary columns (ne30 L30, 2 ranks: ~1.3 MB)"
NCCL_DEBUG=INFO run 2 --halo rccl --steps 3 --warmup 1 --develop 1 --sustained-steps 0 --no-cpu-baseline 2>gpurun_out/first_contact_1.err | tee gpurun_out/first_contact_1.json | \
	field "d['multi_rank']"
grep -c "nranks 2" gpurun_out/first_contact_1.err

echo "== 2. two ranks, --halo auto: the peer-to-peer transport (IPC-mapped ghost buffers written over xGMI) against RCCL"
echo "   expect: config.parallelism says 'identical bytes' and gives both times per DSS; parity true.  If the mapping is refused"
echo "           (no peer access) it says so and continues with RCCL -- that is the designed fallback, not an error."
run 2 --halo auto --steps 10 --warmup 2 --develop 2 --sustained-steps 0 --no-cpu-baseline 2>gpurun_out/first_contact_2.err | tee gpurun_out/first_contact_2.json | \
	field "(d['config']['parallelism'], d['multi_rank']['parity'])"

echo "== 3. strong scaling of the headline problem (ne30 L30), RCCL: N = 1, 2, 4, $NMAX"
echo "   expect (DESIGN.md section 6, projected from the per-rank compute share + 10-30 us exposed per exchange): ms_per_step ~ 1.14-1.17 / 0.91-1.03 / 0.65-0.77 / 0.56-0.68;"
echo "           parity true at every N; early/late tiles non-zero at every N > 1 (the boundary-first split is on)"
for n in 1 2 4 $NMAX; do
	if [ $n -eq 1 ]; then python bench.py --steps 20 --warmup 3 --sustained-steps 0 --no-cpu-baseline > gpurun_out/first_contact_3_n1.json 2>/dev/null;
	else run $n --halo rccl --steps 20 --warmup 3 --sustained-steps 0 --no-cpu-baseline > gpurun_out/first_contact_3_n$n.json 2>gpurun_out/first_contact_3_n$n.err; fi
	field "(d['n_gpus'], round(d['ms_per_step'], 3), d.get('multi_rank', {}).get('parity'))" < gpurun_out/first_contact_3_n$n.json
done

echo "== 4. the same with the peer-to-peer transport (if step 2 accepted it)"
for n in 2 4 $NMAX; do
	run $n --halo p2p --steps 20 --warmup 3 --sustained-steps 0 --no-cpu-baseline > gpurun_out/first_contact_4_n$n.json 2>gpurun_out/first_contact_4_n$n.err || echo "   N=$n: p2p refused (see gpurun_out/first_contact_4_n$n.err)"
	field "(d['n_gpus'], round(d['ms_per_step'], 3), d['multi_rank']['parity'])" < gpurun_out/first_contact_4_n$n.json
done

echo "== 5. weak scaling (the regime the design is for): ne = 30 sqrt(N), dt scaled"
echo "   expect: element-DOF-updates/s growing ~ linearly with N (per-GPU work is fixed; the exchange grows with sqrt(N))"
for n in 2 4 $NMAX; do
	run $n --scaling weak --halo rccl --steps 10 --warmup 2 --sustained-steps 0 --no-cpu-baseline > gpurun_out/first_contact_5_n$n.json 2>gpurun_out/first_contact_5_n$n.err
	field "(d['n_gpus'], d['config']['workload'][:60], round(d['ms_per_step'], 3), '%.3g' % d['value'], d['multi_rank']['parity'])" < gpurun_out/first_contact_5_n$n.json
done
