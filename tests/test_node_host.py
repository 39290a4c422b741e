"""Node.js host mirror (sph-pie_amd/host) over the raw N-API addon.  The CPU leg checks the pure-host logic
against the hand-derived vectors; the GPU leg replays the sessionStore golden vectors through the device-backed
store and exercises GET /api/calendar end to end."""
import os
import shutil
import subprocess

import pytest

from conftest import REPO

HOST = os.path.join(REPO, "sph-pie_amd", "host")
node = shutil.which("node")
needs_node = pytest.mark.skipif(node is None, reason="node is not installed on this machine")


def run_node(script, timeout):
    env = dict(os.environ, TZ="UTC")
    res = subprocess.run([node, os.path.join(HOST, "test", script)], cwd=REPO, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout
    return res.stdout


@needs_node
def test_addon_builds_and_host_logic_cpu(pie):
    assert pie.build_napi() is not None, "node headers (node_api.h) not found"
    out = run_node("cpu_test.js", 120)
    assert "host cpu_test ok" in out


@needs_node
def test_host_cutoff_matches_js_date_vectors_in_every_zone():
    """a11: host/calendarFeed.js getCalendarCutoffTimestamp under each of the seven zones of tests/golden/cutoff_zones.json
    (one node process per TZ: Node 12 reads TZ once) gives the vectors' values — the same vectors that pin the Python
    restatement in test_oracle_golden.py."""
    import json
    doc = json.load(open(os.path.join(REPO, "tests", "golden", "cutoff_zones.json")))
    script = ("const cf=require('./sph-pie_amd/host/calendarFeed.js');"
              "const z=JSON.parse(require('fs').readFileSync('tests/golden/cutoff_zones.json','utf8')).zones[process.env.TZ];"
              "let bad=0;for(const [now,back,want] of z){if(cf.getCalendarCutoffTimestamp(back,now)!==want){bad++;}}"
              "console.log(z.length+' '+bad)")
    for tz in doc["zones"]:
        res = subprocess.run([node, "-e", script], cwd=REPO, env=dict(os.environ, TZ=tz), stdout=subprocess.PIPE, text=True,
                             timeout=60, check=True)
        n, bad = res.stdout.split()
        assert int(n) == len(doc["zones"][tz]) and int(bad) == 0, (tz, res.stdout)


@needs_node
def test_reference_faithful_js_matches_c_oracle(oracle):
    """The JS restatement (bench baseline B1 + checker of the Node GPU test) agrees with the C oracle."""
    script = ("const r=require('./oracle/ref_faithful.js');const rows=r.genCorpus(0x5EED5EEDn,1000,10,3);"
              "const T0=1700000000000;const f=r.scanFeeds(r.buildMap(rows),10,T0-100*86400000,T0-110*86400000,d=>d!==1);"
              "console.log(JSON.stringify({rows:rows.map(x=>[x.start,x.end,x.user,x.disc]),feeds:f.map(x=>x.map(y=>y.row))}))")
    res = subprocess.run([node, "-e", script], cwd=REPO, stdout=subprocess.PIPE, text=True, timeout=120, check=True)
    import json
    import numpy as np
    doc = json.loads(res.stdout)
    s, e, u, d = oracle.gen(0x5EED5EED, 1000, 0, 1000, 10, 3, 0)
    assert np.array_equal(np.array(doc["rows"]), np.stack([s, e, u, d], 1))
    T0 = 1700000000000
    c, o, idx = oracle.scan(s, e, u, d, 10, T0 - 100 * 86400000, T0 - 110 * 86400000, 0b101)
    assert [len(f) for f in doc["feeds"]] == c.tolist()
    assert sum(doc["feeds"], []) == idx.tolist() and len(idx) > 0


@needs_node
@pytest.mark.gpu
def test_node_host_on_gpu(pie):
    assert pie.build_napi() is not None
    out = run_node("gpu_test.js", 300)
    assert "host gpu_test ok" in out


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("zone", ["America/New_York", "Australia/Lord_Howe", "America/Havana", "UTC"])
def test_retention_purge_through_the_node_host_in_a_dst_zone(pie, zone):
    """f2 through the Node host under a real time zone: store.purgeRetention (device month arithmetic under the table
    host/tzTable.js builds from the engine's zone rules) drops exactly the sessions the JS engine's own Date arithmetic — the
    reference's, sqlProvider.js:991-1009 — says, incl. sessions whose shifted instant lands on a clock change."""
    assert pie.build_napi() is not None
    env = dict(os.environ, TZ=zone)
    res = subprocess.run([node, os.path.join(HOST, "test", "gpu_tz_test.js")], cwd=REPO, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0, res.stdout
    assert "host gpu_tz_test ok" in res.stdout
