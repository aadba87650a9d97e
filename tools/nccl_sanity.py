import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", world_size=1, rank=0)
x = torch.arange(1000003, dtype=torch.float32, device="cuda:0")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    y = x * 2
torch.cuda.current_stream().wait_stream(side)
h1 = dist.all_reduce(y[123457:], op=dist.ReduceOp.SUM, async_op=True)
h2 = dist.all_reduce(y[:123457], op=dist.ReduceOp.SUM, async_op=True)
h1.wait(); h2.wait()
torch.cuda.synchronize()
print("nccl ok", float(y[-1]), float(y[5]))
dist.destroy_process_group()
