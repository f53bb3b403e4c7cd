import json,sys
d=json.load(open(sys.argv[1]))
print(d['workload'], d['n'], d['nnz'], d['one_gpu_ms'], 'call_us', round(d['exchange_model']['us_per_call'],1))
for w,rec in d['per_world'].items():
    print('P',w,'sum_slowest',round(rec['sum_of_slowest_ranks_ms'],3),'exch',round(rec['exchange_ms'],3),'step',round(rec['projected_step_ms'],3),'bench_step',round(rec['projected_bench_step_ms'],3),'speedup',round(rec['projected_speedup'],2))
    for ph,v in rec['phases'].items():
        print('   ',ph,{k:round(x,3) for k,x in v.items()})
    for rule,v in rec['egonet_cut_rules'].items():
        print('    ego',rule,round(v['max_ms'],3),round(v['imbalance'],2))
