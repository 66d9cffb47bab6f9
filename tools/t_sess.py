import time, sys
sys.path.insert(0, '/root/repo')
t0=time.time()
import fvdb_import
fv = fvdb_import.load()
print("import", time.time()-t0); t0=time.time()
ctx = fv.Context(0); print("ctx", time.time()-t0); t0=time.time()
s = fv.VectorDbSession(ctx); print("session", time.time()-t0); t0=time.time()
s.add_vectors([{"id": f"doc-{i}", "vector": [float(i), 1.0, 0.5], "metadata": {"n": i}} for i in range(20)])
print("add 20", time.time()-t0); t0=time.time()
for k in (3, 10, 100):
    s.search([0.0, 1.0, 0.5], k); print("search", k, time.time()-t0); t0=time.time()
