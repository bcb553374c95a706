#!/bin/bash
# Staging chunk size and copy threads of the pipe sink (one shard, one pipe): gpurun_out/r03_pipe_chunks.log
for c in 4 8 16 32 64; do
  echo "chunk $c MiB"
  AUR_PIPE_CHUNK_MB=$c timeout -k 10 120 python tools/pipe_shards.py --shards 1 --frames 1920 --threads 4 2>&1 | grep shards || exit 1
done
for t in 1 2 3 6; do
  echo "threads $t"
  timeout -k 10 120 python tools/pipe_shards.py --shards 1 --frames 1920 --threads $t 2>&1 | grep shards || exit 1
done
