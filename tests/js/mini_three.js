// mini_three.js -- the two closed forms the addon tests need to pose a camera, written out for column-major
// {elements[16]} matrices (three.js r147: Matrix4.compose for a yaw about +y with unit scale, PerspectiveCamera's projection).
// Test infrastructure; the product takes camera matrices as inputs and never builds them.
'use strict';
function composeYaw(p, yawDeg) {
  const h = yawDeg * Math.PI / 360, y = Math.sin(h), w = Math.cos(h);
  const y2 = y + y, yy = y * y2, wy = w * y2;
  return { elements: [1 - yy, 0, -wy, 0, 0, 1, 0, 0, wy, 0, 1 - yy, 0, p[0], p[1], p[2], 1] };
}
function perspective(fovDeg, aspect, near, far) {
  const top = near * Math.tan(fovDeg * Math.PI / 360), height = 2 * top, width = aspect * height, left = -0.5 * width;
  const right = left + width, bottom = top - height;
  const x = 2 * near / (right - left), yv = 2 * near / (top - bottom), a = (right + left) / (right - left), b = (top + bottom) / (top - bottom);
  const c = -(far + near) / (far - near), d = -2 * far * near / (far - near);
  return { elements: [x, 0, 0, 0, 0, yv, 0, 0, a, b, c, -1, 0, 0, d, 0] };
}
module.exports = { composeYaw, perspective };
