cd /root/repo
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests -m gpu -q -k "channel_shard or sharded or marginal or two_rank or poll_sequence" > gpurun_out/r2d/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2d/pytest.log
tail -30 gpurun_out/r2d/pytest.log
