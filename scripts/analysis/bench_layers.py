import json,sys
d=json.load(open(sys.argv[1]))
l={x['kernel']:x['launch_ms'] for x in d['roofline_mfma']}
print(sys.argv[2], d['value'], 'enc', d['stage_ms']['sp_gpu_infer:encoder'], 'heads', d['stage_ms']['sp_gpu_infer:heads'], {k:l[k] for k in l if k.startswith(('conv3b','conv4','convPa'))})
