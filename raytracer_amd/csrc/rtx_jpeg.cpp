// JPEG decoding for textures (baseline and progressive Huffman, 8-bit, 1/3/4 components, restart
// intervals) to 8-bit RGB, top row first.
//
// The reference loads textures through its vendored stb_image (src/Texture.cpp:4-8, three channels forced).
// A JPEG decoder's output depends on its arithmetic, so the numeric stages below restate stb_image's:
// the 13-bit fixed-point inverse DCT with its rounding points (include/stb_image.h:2235-2333), the
// triangle-filter chroma upsampling (:3240-3299, :3417-3425, row selection :3685-3699), the 20-bit fixed-point
// YCbCr conversion (:3429-3452), the RGB / CMYK / YCCK rules (:3642, :3701-3747) and the 16-bit wrap of
// dequantised coefficients (:2034,2051,2066,2884).  The entropy decoding follows ITU-T T.81 (Annex F and G);
// for a well-formed file there is only one answer there.  Corrupt files fail to load.

#include <cstdint>
#include <cstring>
#include <vector>

#include "rtx_image.h"

namespace rtxh {

namespace {

constexpr uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
								41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
								30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
	bool present = false;
	int first[18];       // first code of each length (1..16), left-aligned comparison not needed
	int count[17];
	int offset[17];      // index of the first symbol of each length
	uint8_t symbols[256];
	int16_t fast[512];   // 9 leading bits -> (length << 8) | symbol, or -1
	bool build(const int counts[16], const uint8_t* syms, int n) {
		int code = 0, k = 0;
		for (int len = 1; len <= 16; ++len) {
			count[len] = counts[len - 1];
			first[len] = code;
			offset[len] = k;
			code += count[len];
			if (count[len] && code - 1 >= (1 << len)) return false;  // more codes than the length can hold
			k += count[len];
			code <<= 1;
		}
		if (k != n || n > 256) return false;
		std::memcpy(symbols, syms, (size_t)n);
		for (int i = 0; i < 512; ++i) fast[i] = -1;
		for (int len = 1; len <= 9; ++len)
			for (int c = 0; c < count[len]; ++c) {
				const int codeword = first[len] + c;
				const int lo = codeword << (9 - len);
				for (int j = 0; j < (1 << (9 - len)); ++j) fast[lo + j] = (int16_t)((len << 8) | symbols[offset[len] + c]);
			}
		present = true;
		return true;
	}
};

struct Component {
	int id = 0, h = 1, v = 1, tq = 0;
	int hd = 0, ha = 0;       // Huffman table selectors of the current scan
	int pred = 0;             // DC predictor
	int x = 0, y = 0;         // samples this component really has
	int w2 = 0, h2 = 0;       // padded to whole MCUs
	std::vector<uint8_t> data;    // w2 * h2 decoded samples
	std::vector<int16_t> coeff;   // progressive: 64 per block, blocks in rows of w2 / 8
};

struct Decoder {
	const uint8_t* p;
	const uint8_t* end;
	// entropy-coded segment reader
	uint32_t bitBuf = 0;
	int bitCnt = 0;
	int marker = -1;     // marker met while reading entropy-coded bytes (-1: none)
	bool noMore = false;
	bool bad = false;

	uint16_t dequant[4][64];
	Huffman dc[4], ac[4];
	Component comp[4];
	int ncomp = 0, width = 0, height = 0;
	int hmax = 1, vmax = 1, mcuX = 0, mcuY = 0;
	bool progressive = false, jfif = false;
	int adobeTransform = -1, rgbIds = 0;
	int restartInterval = 0, todo = 0;
	int scanN = 0, order[4];
	int specStart = 0, specEnd = 63, succHigh = 0, succLow = 0, eobRun = 0;

	int get8() { return p < end ? *p++ : 0; }
	int get16() {
		int a = get8();
		return (a << 8) | get8();
	}
	bool eof() const { return p >= end; }

	// ---- bits.  Bytes are appended below the valid bits; FF 00 is a literal FF, FF followed by anything
	// else is a marker, after which zeros are supplied (T.81 F.2.2.5).
	void fill() {
		while (bitCnt <= 24) {
			int b = 0;
			if (!noMore) {
				b = get8();
				if (b == 0xFF) {
					int c = get8();
					while (c == 0xFF) c = get8();  // fill bytes
					if (c != 0) {
						marker = c;
						noMore = true;
						b = 0;
					}
				}
			}
			bitBuf |= (uint32_t)b << (24 - bitCnt);
			bitCnt += 8;
		}
	}
	int bits(int n) {  // n in 0..16, most significant first
		if (n == 0) return 0;
		if (bitCnt < n) fill();
		const int v = (int)(bitBuf >> (32 - n));
		bitBuf <<= n;
		bitCnt -= n;
		return v;
	}
	int bit() { return bits(1); }
	// receive + extend (T.81 F.2.2.1)
	int receive_extend(int n) {
		if (n == 0) return 0;
		const int v = bits(n);
		return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
	}
	int symbol(const Huffman& h) {
		if (bitCnt < 16) fill();
		const int f = h.fast[bitBuf >> 23];
		if (f >= 0) {
			const int len = f >> 8;
			bitBuf <<= len;
			bitCnt -= len;
			return f & 0xFF;
		}
		int code = (int)(bitBuf >> 23);
		for (int len = 10; len <= 16; ++len) {
			code = (int)(bitBuf >> (32 - len));
			if (code - h.first[len] < h.count[len] && code >= h.first[len]) {
				bitBuf <<= len;
				bitCnt -= len;
				return h.symbols[h.offset[len] + code - h.first[len]];
			}
		}
		bad = true;
		return 0;
	}
	void reset_entropy() {
		bitBuf = 0;
		bitCnt = 0;
		noMore = false;
		marker = -1;
		for (auto& c : comp) c.pred = 0;
		todo = restartInterval ? restartInterval : 0x7FFFFFFF;
		eobRun = 0;
	}
	// After an MCU: at the end of a restart interval the next marker must be RSTn; anything else ends the scan.
	bool interval_done() {
		if (--todo > 0) return true;
		if (bitCnt < 24) fill();
		if (marker < 0xD0 || marker > 0xD7) return false;
		reset_entropy();
		return true;
	}

	// ---- blocks
	void block_baseline(int16_t* d, Component& c) {
		std::memset(d, 0, 128);
		const uint16_t* q = dequant[c.tq];
		const int t = symbol(dc[c.hd]);
		if (bad || t > 16) {
			bad = true;
			return;
		}
		c.pred += receive_extend(t);
		d[0] = (int16_t)(c.pred * q[0]);
		for (int k = 1; k < 64;) {
			const int rs = symbol(ac[c.ha]);
			if (bad) return;
			const int r = rs >> 4, s = rs & 15;
			if (s == 0) {
				if (rs != 0xF0) break;
				k += 16;
				continue;
			}
			k += r;
			if (k > 63) {
				bad = true;
				return;
			}
			const int z = ZIGZAG[k++];
			d[z] = (int16_t)(receive_extend(s) * q[z]);
		}
	}
	void block_dc_progressive(int16_t* d, Component& c) {
		if (specEnd != 0) {
			bad = true;
			return;
		}
		if (succHigh == 0) {
			std::memset(d, 0, 128);
			const int t = symbol(dc[c.hd]);
			if (bad || t > 16) {
				bad = true;
				return;
			}
			c.pred += receive_extend(t);
			d[0] = (int16_t)(c.pred * (1 << succLow));
		}
		else if (bit())
			d[0] = (int16_t)(d[0] + (1 << succLow));
	}
	void refine(int16_t& v, int16_t bitv) {
		if (bit() && (v & bitv) == 0) v = (int16_t)(v > 0 ? v + bitv : v - bitv);
	}
	void block_ac_progressive(int16_t* d, Component& c) {
		if (specStart == 0) {
			bad = true;
			return;
		}
		const Huffman& h = ac[c.ha];
		if (succHigh == 0) {
			if (eobRun) {
				--eobRun;
				return;
			}
			for (int k = specStart; k <= specEnd;) {
				const int rs = symbol(h);
				if (bad) return;
				const int r = rs >> 4, s = rs & 15;
				if (s == 0) {
					if (r < 15) {
						eobRun = (1 << r) - 1;
						if (r) eobRun += bits(r);
						break;
					}
					k += 16;
					continue;
				}
				k += r;
				if (k > 63) {
					bad = true;
					return;
				}
				d[ZIGZAG[k++]] = (int16_t)(receive_extend(s) * (1 << succLow));
			}
			return;
		}
		// refinement of bits already started (T.81 G.1.2.3)
		const int16_t bitv = (int16_t)(1 << succLow);
		if (eobRun) {
			--eobRun;
			for (int k = specStart; k <= specEnd; ++k) {
				int16_t& v = d[ZIGZAG[k]];
				if (v != 0) refine(v, bitv);
			}
			return;
		}
		for (int k = specStart; k <= specEnd;) {
			const int rs = symbol(h);
			if (bad) return;
			int r = rs >> 4;
			int s = rs & 15;
			int value = 0;
			if (s == 0) {
				if (r < 15) {
					eobRun = (1 << r) - 1;
					if (r) eobRun += bits(r);
					r = 64;  // run to the end of the band, refining what is already non-zero
				}
			}
			else {
				if (s != 1) {
					bad = true;
					return;
				}
				value = bit() ? bitv : -bitv;
			}
			while (k <= specEnd) {
				int16_t& v = d[ZIGZAG[k++]];
				if (v != 0) refine(v, bitv);
				else {
					if (r == 0) {
						v = (int16_t)value;
						break;
					}
					--r;
				}
			}
		}
	}
};

// Fixed-point constants of the inverse DCT: (int)(x * 4096 + 0.5) with x a float literal.
constexpr int fx(float x) { return (int)(x * 4096 + 0.5); }

// 32-bit arithmetic that wraps: coefficients of a damaged file can overflow the fixed-point range, and the pixels
// must then be the ones two's-complement hardware gives (the reference's plain int arithmetic in practice).
struct Wrap {
	uint32_t u;
	Wrap(int v = 0) : u((uint32_t)v) {}
	friend Wrap operator+(Wrap a, Wrap b) { Wrap r; r.u = a.u + b.u; return r; }
	friend Wrap operator-(Wrap a, Wrap b) { Wrap r; r.u = a.u - b.u; return r; }
	friend Wrap operator*(Wrap a, Wrap b) { Wrap r; r.u = a.u * b.u; return r; }
	int shr(int k) const { return (int)u >> k; }  // arithmetic shift of the signed value
};

// One 8-point inverse DCT; in: s0..s7; out: the four even-part sums x0..x3 and odd-part terms t0..t3.
struct Idct1 {
	Wrap x0, x1, x2, x3, t0, t1, t2, t3;
	Idct1(Wrap s0, Wrap s1, Wrap s2, Wrap s3, Wrap s4, Wrap s5, Wrap s6, Wrap s7) {
		Wrap p1 = (s2 + s6) * fx(0.5411961f);
		const Wrap e2 = p1 + s6 * fx(-1.847759065f);
		const Wrap e3 = p1 + s2 * fx(0.765366865f);
		const Wrap e0 = (s0 + s4) * 4096;
		const Wrap e1 = (s0 - s4) * 4096;
		x0 = e0 + e3;
		x3 = e0 - e3;
		x1 = e1 + e2;
		x2 = e1 - e2;
		Wrap a0 = s7, a1 = s5, a2 = s3, a3 = s1;
		Wrap p3 = a0 + a2, p4 = a1 + a3;
		p1 = a0 + a3;
		Wrap p2 = a1 + a2;
		const Wrap p5 = (p3 + p4) * fx(1.175875602f);
		a0 = a0 * fx(0.298631336f);
		a1 = a1 * fx(2.053119869f);
		a2 = a2 * fx(3.072711026f);
		a3 = a3 * fx(1.501321110f);
		p1 = p5 + p1 * fx(-0.899976223f);
		p2 = p5 + p2 * fx(-2.562915447f);
		p3 = p3 * fx(-1.961570560f);
		p4 = p4 * fx(-0.390180644f);
		t3 = a3 + p1 + p4;
		t2 = a2 + p2 + p3;
		t1 = a1 + p2 + p4;
		t0 = a0 + p1 + p3;
	}
};

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

void idct_block(uint8_t* out, int stride, const int16_t* d) {
	int val[64];
	for (int i = 0; i < 8; ++i) {  // columns, two extra bits of precision kept
		const Idct1 c(d[i], d[8 + i], d[16 + i], d[24 + i], d[32 + i], d[40 + i], d[48 + i], d[56 + i]);
		const Wrap x0 = c.x0 + 512, x1 = c.x1 + 512, x2 = c.x2 + 512, x3 = c.x3 + 512;
		val[i] = (x0 + c.t3).shr(10);
		val[56 + i] = (x0 - c.t3).shr(10);
		val[8 + i] = (x1 + c.t2).shr(10);
		val[48 + i] = (x1 - c.t2).shr(10);
		val[16 + i] = (x2 + c.t1).shr(10);
		val[40 + i] = (x2 - c.t1).shr(10);
		val[24 + i] = (x3 + c.t0).shr(10);
		val[32 + i] = (x3 - c.t0).shr(10);
	}
	for (int i = 0; i < 8; ++i) {  // rows: remove 2^17, round, and move -128..127 to 0..255
		const int* v = val + i * 8;
		uint8_t* o = out + (size_t)i * stride;
		const Idct1 c(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
		const int bias = 65536 + (128 << 17);
		const Wrap x0 = c.x0 + bias, x1 = c.x1 + bias, x2 = c.x2 + bias, x3 = c.x3 + bias;
		o[0] = clamp8((x0 + c.t3).shr(17));
		o[7] = clamp8((x0 - c.t3).shr(17));
		o[1] = clamp8((x1 + c.t2).shr(17));
		o[6] = clamp8((x1 - c.t2).shr(17));
		o[2] = clamp8((x2 + c.t1).shr(17));
		o[5] = clamp8((x2 - c.t1).shr(17));
		o[3] = clamp8((x3 + c.t0).shr(17));
		o[4] = clamp8((x3 - c.t0).shr(17));
	}
}

bool parse_tables(Decoder& z, int m) {
	if (m == 0xDD) {  // DRI
		if (z.get16() != 4) return false;
		z.restartInterval = z.get16();
		return true;
	}
	if (m == 0xDB) {  // DQT
		int len = z.get16() - 2;
		while (len > 0) {
			const int q = z.get8();
			const int wide = q >> 4, t = q & 15;
			if (wide > 1 || t > 3) return false;
			for (int i = 0; i < 64; ++i) z.dequant[t][ZIGZAG[i]] = (uint16_t)(wide ? z.get16() : z.get8());
			len -= wide ? 129 : 65;
		}
		return len == 0;
	}
	if (m == 0xC4) {  // DHT
		int len = z.get16() - 2;
		while (len > 0) {
			const int q = z.get8();
			const int cls = q >> 4, th = q & 15;
			if (cls > 1 || th > 3) return false;
			int counts[16], n = 0;
			for (int i = 0; i < 16; ++i) {
				counts[i] = z.get8();
				n += counts[i];
			}
			if (n > 256) return false;
			uint8_t syms[256];
			for (int i = 0; i < n; ++i) syms[i] = (uint8_t)z.get8();
			if (!(cls ? z.ac[th] : z.dc[th]).build(counts, syms, n)) return false;
			len -= 17 + n;
		}
		return len == 0;
	}
	if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {  // APPn, COM
		int len = z.get16();
		if (len < 2) return false;
		len -= 2;
		if (m == 0xE0 && len >= 5) {
			const char tag[5] = {'J', 'F', 'I', 'F', 0};
			bool ok = true;
			for (int i = 0; i < 5; ++i) ok &= z.get8() == tag[i];
			len -= 5;
			if (ok) z.jfif = true;
		}
		else if (m == 0xEE && len >= 12) {
			const char tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
			bool ok = true;
			for (int i = 0; i < 6; ++i) ok &= z.get8() == tag[i];
			len -= 6;
			if (ok) {
				z.get8();
				z.get16();
				z.get16();
				z.adobeTransform = z.get8();
				len -= 6;
			}
		}
		if (len > z.end - z.p) return false;
		z.p += len;
		return true;
	}
	return false;
}

// A marker outside entropy-coded data: the one met while decoding, else the next FF xx pair.
int next_marker(Decoder& z) {
	if (z.marker >= 0) {
		const int m = z.marker;
		z.marker = -1;
		return m;
	}
	int x = z.get8();
	if (x != 0xFF) return -1;
	while (x == 0xFF) x = z.get8();
	return x;
}

bool frame_header(Decoder& z) {
	const int len = z.get16();
	if (len < 11 || z.get8() != 8) return false;
	z.height = z.get16();
	z.width = z.get16();
	if (z.height == 0 || z.width == 0) return false;
	z.ncomp = z.get8();
	if (z.ncomp != 1 && z.ncomp != 3 && z.ncomp != 4) return false;
	if (len != 8 + 3 * z.ncomp) return false;
	if ((long long)z.width * z.height > (1ll << 28)) return false;
	z.rgbIds = 0;
	for (int i = 0; i < z.ncomp; ++i) {
		Component& c = z.comp[i];
		c.id = z.get8();
		if (z.ncomp == 3 && c.id == "RGB"[i]) ++z.rgbIds;
		const int q = z.get8();
		c.h = q >> 4;
		c.v = q & 15;
		c.tq = z.get8();
		if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return false;
		z.hmax = c.h > z.hmax ? c.h : z.hmax;
		z.vmax = c.v > z.vmax ? c.v : z.vmax;
	}
	z.mcuX = (z.width + z.hmax * 8 - 1) / (z.hmax * 8);
	z.mcuY = (z.height + z.vmax * 8 - 1) / (z.vmax * 8);
	for (int i = 0; i < z.ncomp; ++i) {
		Component& c = z.comp[i];
		c.x = (z.width * c.h + z.hmax - 1) / z.hmax;
		c.y = (z.height * c.v + z.vmax - 1) / z.vmax;
		c.w2 = z.mcuX * c.h * 8;
		c.h2 = z.mcuY * c.v * 8;
		c.data.assign((size_t)c.w2 * c.h2, 0);
		if (z.progressive) c.coeff.assign((size_t)c.w2 * c.h2, 0);
	}
	return true;
}

bool scan_header(Decoder& z) {
	const int len = z.get16();
	z.scanN = z.get8();
	if (z.scanN < 1 || z.scanN > 4 || z.scanN > z.ncomp || len != 6 + 2 * z.scanN) return false;
	for (int i = 0; i < z.scanN; ++i) {
		const int id = z.get8(), q = z.get8();
		int which = 0;
		while (which < z.ncomp && z.comp[which].id != id) ++which;
		if (which == z.ncomp) return false;
		z.comp[which].hd = q >> 4;
		z.comp[which].ha = q & 15;
		if (z.comp[which].hd > 3 || z.comp[which].ha > 3) return false;
		z.order[i] = which;
	}
	z.specStart = z.get8();
	z.specEnd = z.get8();
	const int a = z.get8();
	z.succHigh = a >> 4;
	z.succLow = a & 15;
	if (z.progressive) {
		if (z.specStart > 63 || z.specEnd > 63 || z.specStart > z.specEnd || z.succHigh > 13 || z.succLow > 13) return false;
	}
	else {
		if (z.specStart != 0 || z.succHigh != 0 || z.succLow != 0) return false;
		z.specEnd = 63;
	}
	return true;
}

// One block of component c at block column bx, block row by.
void decode_one(Decoder& z, Component& c, int bx, int by) {
	if (!z.progressive) {
		int16_t d[64];
		z.block_baseline(d, c);
		if (!z.bad) idct_block(c.data.data() + (size_t)c.w2 * by * 8 + (size_t)bx * 8, c.w2, d);
		return;
	}
	int16_t* d = c.coeff.data() + 64 * ((size_t)bx + (size_t)by * (c.w2 / 8));
	if (z.specStart == 0) z.block_dc_progressive(d, c);
	else z.block_ac_progressive(d, c);
}

bool entropy_coded_data(Decoder& z) {
	z.reset_entropy();
	for (int i = 0; i < z.scanN; ++i) {
		const Component& c = z.comp[z.order[i]];
		if ((z.progressive ? z.specStart == 0 : true) && !z.dc[c.hd].present && !(z.progressive && z.succHigh)) return false;
		if ((!z.progressive || z.specStart != 0) && !z.ac[c.ha].present) return false;
	}
	if (z.scanN == 1) {
		// a scan of one component covers just the blocks that component really has
		Component& c = z.comp[z.order[0]];
		const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
		for (int by = 0; by < h; ++by)
			for (int bx = 0; bx < w; ++bx) {
				decode_one(z, c, bx, by);
				if (z.bad) return false;
				if (!z.interval_done()) return true;
			}
		return true;
	}
	if (z.progressive && z.specStart != 0) return false;  // AC scans hold one component
	for (int my = 0; my < z.mcuY; ++my)
		for (int mx = 0; mx < z.mcuX; ++mx) {
			for (int k = 0; k < z.scanN; ++k) {
				Component& c = z.comp[z.order[k]];
				for (int y = 0; y < c.v; ++y)
					for (int x = 0; x < c.h; ++x) {
						decode_one(z, c, mx * c.h + x, my * c.v + y);
						if (z.bad) return false;
					}
			}
			if (!z.interval_done()) return true;
		}
	return true;
}

void finish_progressive(Decoder& z) {
	for (int n = 0; n < z.ncomp; ++n) {
		Component& c = z.comp[n];
		const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
		for (int by = 0; by < h; ++by)
			for (int bx = 0; bx < w; ++bx) {
				int16_t* d = c.coeff.data() + 64 * ((size_t)bx + (size_t)by * (c.w2 / 8));
				for (int i = 0; i < 64; ++i) d[i] = (int16_t)(d[i] * z.dequant[c.tq][i]);
				idct_block(c.data.data() + (size_t)c.w2 * by * 8 + (size_t)bx * 8, c.w2, d);
			}
	}
}

// ---- upsampling: one output row of `w` low-resolution samples widened by hs, from the nearer and the
// farther of the two source rows
void upsample_row(uint8_t* out, const uint8_t* nearRow, const uint8_t* farRow, int w, int hs, int vs) {
	if (hs == 1 && vs == 1) {
		std::memcpy(out, nearRow, (size_t)w);
	}
	else if (hs == 1 && vs == 2) {
		for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * nearRow[i] + farRow[i] + 2) >> 2);
	}
	else if (hs == 2 && vs == 1) {
		const uint8_t* in = nearRow;
		if (w == 1) {
			out[0] = out[1] = in[0];
			return;
		}
		out[0] = in[0];
		out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
		int i;
		for (i = 1; i < w - 1; ++i) {
			const int n = 3 * in[i] + 2;
			out[i * 2] = (uint8_t)((n + in[i - 1]) >> 2);
			out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
		}
		out[i * 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
		out[i * 2 + 1] = in[w - 1];
	}
	else if (hs == 2 && vs == 2) {
		if (w == 1) {
			out[0] = out[1] = (uint8_t)((3 * nearRow[0] + farRow[0] + 2) >> 2);
			return;
		}
		int t1 = 3 * nearRow[0] + farRow[0];
		out[0] = (uint8_t)((t1 + 2) >> 2);
		for (int i = 1; i < w; ++i) {
			const int t0 = t1;
			t1 = 3 * nearRow[i] + farRow[i];
			out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
			out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
		}
		out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
	}
	else {
		for (int i = 0; i < w; ++i)
			for (int j = 0; j < hs; ++j) out[i * hs + j] = nearRow[i];
	}
}

// 20-bit fixed point: ((int)(x * 4096.0f + 0.5f)) << 8
constexpr int fx20(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }

inline void ycc_to_rgb(uint8_t* out, int y, int cbv, int crv) {
	const int yf = (y << 20) + (1 << 19);
	const int cr = crv - 128, cb = cbv - 128;
	int r = yf + cr * fx20(1.40200f);
	int g = yf + cr * -fx20(0.71414f) + (int)((unsigned)(cb * -fx20(0.34414f)) & 0xffff0000u);
	int b = yf + cb * fx20(1.77200f);
	out[0] = clamp8(r >> 20);
	out[1] = clamp8(g >> 20);
	out[2] = clamp8(b >> 20);
}

inline uint8_t mul8(int x, int y) {
	const unsigned t = (unsigned)(x * y + 128);
	return (uint8_t)((t + (t >> 8)) >> 8);
}

}  // namespace

bool load_jpeg(const std::vector<uint8_t>& f, Image& img) {
	if (f.size() < 4 || f[0] != 0xFF || f[1] != 0xD8) return false;
	Decoder z;
	z.p = f.data();
	z.end = f.data() + f.size();
	std::memset(z.dequant, 0, sizeof(z.dequant));
	if (next_marker(z) != 0xD8) return false;
	int m = next_marker(z);
	while (m != 0xC0 && m != 0xC1 && m != 0xC2) {
		if (m < 0 || !parse_tables(z, m)) return false;
		m = next_marker(z);
		while (m < 0) {  // padding after a segment
			if (z.eof()) return false;
			m = next_marker(z);
		}
	}
	z.progressive = m == 0xC2;
	if (!frame_header(z)) return false;
	m = next_marker(z);
	while (m != 0xD9) {
		if (m == 0xDA) {
			if (!scan_header(z) || !entropy_coded_data(z)) return false;
			if (z.marker < 0) {
				// no marker met yet: look for the next FF and take the byte behind it
				while (!z.eof()) {
					if (z.get8() == 0xFF) {
						z.marker = z.get8();
						break;
					}
				}
			}
		}
		else if (m == 0xDC) {  // DNL
			if (z.get16() != 4 || z.get16() != z.height) return false;
		}
		else if (m < 0 || !parse_tables(z, m))
			return false;
		m = next_marker(z);
	}
	if (z.progressive) finish_progressive(z);

	// upsample and convert
	const int n = z.ncomp;
	const bool isRgb = n == 3 && (z.rgbIds == 3 || (z.adobeTransform == 0 && !z.jfif));
	img.w = z.width;
	img.h = z.height;
	img.rgb.assign((size_t)z.width * z.height * 3, 0);
	std::vector<std::vector<uint8_t>> line(n, std::vector<uint8_t>((size_t)z.width + 8 + 3 * 8));
	struct Up {
		int hs, vs, ystep, wLores, ypos;
		const uint8_t *line0, *line1;
	} up[4];
	for (int k = 0; k < n; ++k) {
		const Component& c = z.comp[k];
		up[k].hs = z.hmax / c.h;
		up[k].vs = z.vmax / c.v;
		up[k].ystep = up[k].vs >> 1;
		up[k].wLores = (z.width + up[k].hs - 1) / up[k].hs;
		up[k].ypos = 0;
		up[k].line0 = up[k].line1 = c.data.data();
		line[k].assign((size_t)up[k].wLores * up[k].hs + 8, 0);
	}
	for (int j = 0; j < z.height; ++j) {
		const uint8_t* row[4] = {nullptr, nullptr, nullptr, nullptr};
		for (int k = 0; k < n; ++k) {
			Up& u = up[k];
			const bool bottom = u.ystep >= (u.vs >> 1);
			upsample_row(line[k].data(), bottom ? u.line1 : u.line0, bottom ? u.line0 : u.line1, u.wLores, u.hs, u.vs);
			row[k] = line[k].data();
			if (++u.ystep >= u.vs) {
				u.ystep = 0;
				u.line0 = u.line1;
				if (++u.ypos < z.comp[k].y) u.line1 += z.comp[k].w2;
			}
		}
		uint8_t* out = &img.rgb[(size_t)j * z.width * 3];
		for (int i = 0; i < z.width; ++i, out += 3) {
			if (n == 1) out[0] = out[1] = out[2] = row[0][i];
			else if (n == 3) {
				if (isRgb) {
					out[0] = row[0][i];
					out[1] = row[1][i];
					out[2] = row[2][i];
				}
				else ycc_to_rgb(out, row[0][i], row[1][i], row[2][i]);
			}
			else if (z.adobeTransform == 0) {  // CMYK
				const int k4 = row[3][i];
				out[0] = mul8(row[0][i], k4);
				out[1] = mul8(row[1][i], k4);
				out[2] = mul8(row[2][i], k4);
			}
			else if (z.adobeTransform == 2) {  // YCCK
				ycc_to_rgb(out, row[0][i], row[1][i], row[2][i]);
				const int k4 = row[3][i];
				out[0] = mul8(255 - out[0], k4);
				out[1] = mul8(255 - out[1], k4);
				out[2] = mul8(255 - out[2], k4);
			}
			else ycc_to_rgb(out, row[0][i], row[1][i], row[2][i]);
		}
	}
	return true;
}

}  // namespace rtxh
