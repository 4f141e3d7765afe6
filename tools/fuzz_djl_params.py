#!/usr/bin/env python3
"""Mutation sweep of the DJL .params reader on the host, under AddressSanitizer + UBSan:
  g++ -O1 -g -std=c++17 -fsanitize=address,undefined -Iinclude muzero_amd/host/{djl_params_tool,djl_params,muzero_host}.cpp \
      -o /tmp/asan/djl_tool -Lmuzero_amd -lmuzero_hip -Wl,-rpath,$PWD/muzero_amd
  python3 tools/fuzz_djl_params.py
A valid TicTacToe checkpoint is written, then 300 corrupted copies (random bytes in the headers and records, truncations,
bytes behind an NDAR magic) are read back: every one must either load or be refused with an error - no sanitizer report."""
import subprocess, numpy as np, os, sys, random
sys.path.insert(0,'/root/repo')
from muzero_amd import netspec
tool='/tmp/asan/djl_tool'
env=dict(os.environ, ASAN_OPTIONS='detect_leaks=0')
spec=netspec.SPECS['tictactoe']()
blob=netspec.pack_blob(spec, netspec.random_weights(spec, seed=3))
blob.tofile('/tmp/asan/blob.bin')
r=subprocess.run([tool,'write','tictactoe','/tmp/asan/blob.bin','/tmp/asan','m','1'],capture_output=True,text=True,env=env)
print('write', r.returncode, r.stdout.strip(), r.stderr[-300:])
path=r.stdout.strip()
raw=open(path,'rb').read()
r=subprocess.run([tool,'read','tictactoe',path,'/tmp/asan/back.bin'],capture_output=True,text=True,env=env)
print('read', r.returncode, r.stdout.strip()[:80], r.stderr[-300:])
rng=random.Random(1)
bad=0; crashes=0
for i in range(300):
    b=bytearray(raw)
    mode=rng.randrange(3)
    if mode==0:
        for _ in range(rng.randrange(1,8)):
            pos=rng.randrange(0,min(len(b),4096)) if rng.random()<0.7 else rng.randrange(len(b))
            b[pos]=rng.randrange(256)
    elif mode==1:
        b=b[:rng.randrange(1,len(b))]
    else:
        pos=raw.find(b'NDAR', rng.randrange(0,len(raw)//2))
        if pos>0:
            for k in range(rng.randrange(1,6)): b[pos+4+rng.randrange(0,40)]=rng.randrange(256)
    open('/tmp/asan/mut.params','wb').write(bytes(b))
    r=subprocess.run([tool,'read','tictactoe','/tmp/asan/mut.params','/tmp/asan/x.bin'],capture_output=True,timeout=120,env=env); r.stderr=r.stderr.decode('utf-8','replace'); r.stdout=r.stdout.decode('utf-8','replace')
    if 'ERROR: AddressSanitizer' in r.stderr or 'runtime error' in r.stderr or r.returncode not in (0,1,2):
        crashes+=1
        print('PROBLEM case',i,'mode',mode,'rc',r.returncode, r.stderr[-700:])
        if crashes>3: break
    elif r.returncode!=0: bad+=1
print('mutations rejected', bad, 'of 300; problems', crashes)
