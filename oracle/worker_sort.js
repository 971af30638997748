// TEST INFRASTRUCTURE -- JavaScript restatement of the reference worker's sort (index.js:507-570, semantics in
// SURVEY.md A.1), written from that description; it is the "JS-worker" CPU baseline that bench.py times under node on
// the GPU box's host cores, and tests/test_oracle_golden.py pins it against the vectors the reference's own worker
// produced (oracle/gen_golden.js).  Never part of the product path.
//
//   node oracle/worker_sort.js check <golden dir>
//       every "sort" case of <golden dir>/manifest.json -> one JSON line {cases, failures}
//   node oracle/worker_sort.js bench <rows4.f32> <uniforms.f32> <reps>
//       rows4 = N x (x, y, z, size) f32; uniforms = view[4] (+ cutout[16]); the rows are expanded to the worker's
//       64-byte `matrices` stride before timing (that stride is part of what the reference's sort costs)
//       -> one JSON line {n, kept, ms_median, ms_min, order_sum}
'use strict';
const fs = require('fs');
const path = require('path');

const BUCKETS = 65536;

// One pass of the worker: depth key + culls, order-preserving compaction, 16-bit counting sort (farthest first).
// `m` holds 16 floats per splat of which 12..15 = (x, y, z, size); returns Uint32Array(kept).
function workerSort(m, view, cutout) {
  const total = (m.length / 16) | 0;
  const keptDepth = new Float32Array(total);       // f32 storage of the f64 depth, like the worker's depthList
  const keptIndex = new Uint32Array(total);
  const v0 = view[0], v1 = view[1], v2 = view[2], v3 = view[3];
  let kept = 0, lo = Infinity, hi = -Infinity;
  for (let i = 0, o = 12; i < total; i++, o += 16) {
    const x = m[o], y = m[o + 1], z = m[o + 2];
    const d = v0 * x + v1 * y + v2 * z + v3;         // f64, left to right
    if (!(d < 0) || !(m[o + 3] > -0.0001 * d)) continue;
    if (cutout !== undefined && cutout !== null) {
      const ny = -y;
      const w = 1 / (cutout[3] * x + cutout[7] * ny + cutout[11] * z + cutout[15]);
      let outside = false;
      for (let k = 0; k < 3 && !outside; k++) {
        const q = (cutout[k] * x + cutout[k + 4] * ny + cutout[k + 8] * z + cutout[k + 12]) * w;
        outside = q < -0.5 || q > 0.5;               // NaN -> inside
      }
      if (outside) continue;
    }
    keptDepth[kept] = d;
    keptIndex[kept] = i;
    kept++;
    if (d > hi) hi = d;                              // extremes of the UNROUNDED depths
    if (d < lo) lo = d;
  }
  const scale = (BUCKETS - 1) / (hi - lo);
  const bucketOf = new Int32Array(kept);
  const fill = new Uint32Array(BUCKETS + 1);
  for (let j = 0; j < kept; j++) {
    const b = ((keptDepth[j] - lo) * scale) | 0;     // ToInt32: truncation, NaN -> 0
    bucketOf[j] = b;
    if (b >= 0 && b < BUCKETS) fill[b + 1]++;        // a bucket outside the table is dropped, as in the worker
  }
  for (let b = 0; b < BUCKETS; b++) fill[b + 1] += fill[b];
  const order = new Uint32Array(kept);
  for (let j = 0; j < kept; j++) {
    const b = bucketOf[j];
    if (b >= 0 && b < BUCKETS) order[fill[b]++] = keptIndex[j];
  }
  return order;
}

function expandRows(rows4) {
  const n = (rows4.length / 4) | 0;
  const m = new Float32Array(n * 16);
  for (let i = 0; i < n; i++) { m[16 * i + 12] = rows4[4 * i]; m[16 * i + 13] = rows4[4 * i + 1]; m[16 * i + 14] = rows4[4 * i + 2]; m[16 * i + 15] = rows4[4 * i + 3]; }
  return m;
}

function f32File(file) {
  const b = fs.readFileSync(file);
  return new Float32Array(b.buffer.slice(b.byteOffset, b.byteOffset + (b.length & ~3)));
}

// position-sensitive checksum, sum of value * (position + 1) mod 2^32 (numpy restates it in one vectorised line)
function orderSum(u32) {
  let h = 0;
  for (let i = 0; i < u32.length; i++) h = (h + Math.imul(u32[i], i + 1)) >>> 0;
  return h >>> 0;
}

function check(dir) {
  const manifest = JSON.parse(fs.readFileSync(path.join(dir, 'manifest.json'), 'utf8'));
  const failures = [];
  let cases = 0;
  for (const name of Object.keys(manifest)) {
    const c = manifest[name];
    if (c.kind !== 'sort' || !c.arrays.rows4) continue;
    const raw = fs.readFileSync(path.join(dir, name + '.bin'));
    const arr = (a, T) => new T(raw.buffer.slice(raw.byteOffset + a.offset, raw.byteOffset + a.offset + a.count * 4));
    const rows4 = arr(c.arrays.rows4, Float32Array), view = arr(c.arrays.view, Float32Array);
    const cutout = c.arrays.cutout ? arr(c.arrays.cutout, Float32Array) : undefined;
    const want = arr(c.arrays.sorted, Uint32Array);
    const got = workerSort(expandRows(rows4), view, cutout);
    cases++;
    let ok = got.length === want.length;
    for (let i = 0; ok && i < got.length; i++) ok = got[i] === want[i];
    if (!ok) failures.push(name);
  }
  console.log(JSON.stringify({ cases, failures }));
  process.exit(failures.length ? 1 : 0);
}

function bench(rowsFile, uniformsFile, reps) {
  const m = expandRows(f32File(rowsFile));
  const un = f32File(uniformsFile);
  const view = un.subarray(0, 4), cutout = un.length >= 20 ? un.subarray(4, 20) : undefined;
  const times = [];
  let out = workerSort(m, view, cutout);             // warm-up (JIT)
  for (let r = 0; r < reps; r++) {
    const t = process.hrtime.bigint();
    out = workerSort(m, view, cutout);
    times.push(Number(process.hrtime.bigint() - t) / 1e6);
  }
  times.sort((a, b) => a - b);
  console.log(JSON.stringify({ n: m.length / 16, kept: out.length, ms_median: times[times.length >> 1], ms_min: times[0], order_sum: orderSum(out) }));
}

if (require.main === module) {
  const [mode, a, b, c] = process.argv.slice(2);
  if (mode === 'check') check(a);
  else if (mode === 'bench') bench(a, b, parseInt(c || '5', 10));
  else { console.error('usage: worker_sort.js check <golden dir> | bench <rows4.f32> <uniforms.f32> <reps>'); process.exit(2); }
}
module.exports = { workerSort, expandRows, orderSum };
