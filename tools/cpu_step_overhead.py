"""Host-side enqueue cost per bench step in the distributed flow (1-rank nccl group): accumulate + finish, + async all_reduce,
+ wait and second finalize on a side stream.  python tools/cpu_step_overhead.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29519"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
import torch, torch.distributed as dist
import exblas_amd as ex
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
ex.load_library().exblas_hip_init(-1)
n = 1 << 28
x = ex.gen_dev("ill_cond", n, 1, 1e32)
recs = [ex.new_record_buffer() for _ in range(4)]
side = torch.cuda.Stream()
def t(label, fn, reps=200):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(reps): fn(i)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"{label}: cpu enqueue {1e6*(t1-t0)/reps:.1f} us/step, total {1e6*(t2-t0)/reps:.1f} us/step", flush=True)
def acc_only(i):
    ex.exsum_accumulate_dev(x, 8, True); ex.finish_dev(out=recs[i%4])
t("accumulate+finish", acc_only)
def with_ar(i):
    r=recs[i%4]
    ex.exsum_accumulate_dev(x, 8, True); ex.finish_dev(out=r)
    w=dist.all_reduce(r[ex.OUT_DIGITS:ex.OUT_DIGITS+ex.SET_WORDS], op=dist.ReduceOp.SUM, async_op=True)
    return w
t("+all_reduce async (not waited)", with_ar)
def full(i):
    r=recs[i%4]
    ex.exsum_accumulate_dev(x, 8, True); ex.finish_dev(out=r)
    w=dist.all_reduce(r[ex.OUT_DIGITS:ex.OUT_DIGITS+ex.SET_WORDS], op=dist.ReduceOp.SUM, async_op=True)
    with torch.cuda.stream(side):
        w.wait()
        ex.finalize_dev(r[ex.OUT_DIGITS:ex.OUT_DIGITS+ex.SET_WORDS], out=r)
        ev=torch.cuda.Event(); ev.record()
    torch.cuda.current_stream().wait_event(ev)
t("+wait+finalize2 on side stream", full)
def only_ar(i):
    r=recs[i%4]
    w=dist.all_reduce(r[ex.OUT_DIGITS:ex.OUT_DIGITS+ex.SET_WORDS], op=dist.ReduceOp.SUM, async_op=True)
t("all_reduce call alone", only_ar)
