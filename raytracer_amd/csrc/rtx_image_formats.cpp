// The less common texture containers the reference's image loader accepts (include/stb_image.h, called from
// src/Texture.cpp:6-7 with three channels forced): every BMP flavour short of RLE, the first frame of a GIF,
// flattened RGB PSD, Softimage PIC and Radiance HDR (tone-mapped to 8 bits).  Each decoder returns top-down RGB;
// alpha is dropped the way the reference's 4 -> 3 channel conversion drops it.
// Reads beyond the end of the file yield zeros, as they do in the reference's byte reader (stb_image.h:1490-1501),
// so that short files decode to the same pixels instead of failing.

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rtx_image.h"

namespace rtxh {

namespace {

struct Bytes {
	const uint8_t* p;
	size_t n, pos = 0;
	explicit Bytes(const std::vector<uint8_t>& f) : p(f.data()), n(f.size()) {}
	bool eof() const { return pos >= n; }
	int u8() { return pos < n ? p[pos++] : 0; }
	int le16() {
		int a = u8();
		return a | (u8() << 8);
	}
	uint32_t le32() {
		uint32_t a = (uint32_t)le16();
		return a | ((uint32_t)le16() << 16);
	}
	int be16() {
		int a = u8();
		return (a << 8) | u8();
	}
	uint32_t be32() {
		uint32_t a = (uint32_t)be16();
		return (a << 16) | (uint32_t)be16();
	}
	void skip(long long k) {
		if (k < 0 || (unsigned long long)k > n - pos) pos = n;
		else pos += (size_t)k;
	}
};

bool size_ok(long long w, long long h) { return w > 0 && h > 0 && w <= (1 << 24) && h <= (1 << 24) && w * h <= (1ll << 28); }

// ---- BMP ---------------------------------------------------------------------------------------------------------
// Header versions 12/40/56/108/124, 1/4/8 bits with a colour table, 16/32 bits through channel masks, 24 bits;
// RLE is refused (stb_image.h:5106-5193 header, 5196-5375 pixels).

int top_bit(uint32_t m) {
	int b = -1;
	while (m) {
		++b;
		m >>= 1;
	}
	return b;
}

int pop_count(uint32_t m) {
	int c = 0;
	for (; m; m &= m - 1) ++c;
	return c;
}

// The masked field of a pixel as 8 bits: the field's own bits (at most 8, counted by `bits`) are repeated below
// themselves until a byte is full, so that all-ones maps to 255 (stb_image.h:5081-5098, there as a multiply table).
int widen_field(uint32_t v, int shift, int bits) {
	v = shift < 0 ? v << -shift : v >> shift;  // the field's top bit now sits at bit 7
	v >>= 8 - bits;
	uint32_t acc = 0;
	int have = 0;
	while (have < 8) {
		acc = (acc << bits) | v;
		have += bits;
	}
	return (int)(acc >> (have - 8));
}

}  // namespace

bool is_bmp(const std::vector<uint8_t>& f) {
	if (f.size() < 18 || f[0] != 'B' || f[1] != 'M') return false;
	const uint32_t hsz = (uint32_t)f[14] | (f[15] << 8) | (f[16] << 16) | ((uint32_t)f[17] << 24);
	return hsz == 12 || hsz == 40 || hsz == 56 || hsz == 108 || hsz == 124;
}

bool load_bmp(const std::vector<uint8_t>& f, Image& img) {
	if (!is_bmp(f)) return false;
	Bytes s(f);
	s.skip(10);
	const int offset = (int)s.le32();
	const int hsz = (int)s.le32();
	int consumed = 14;  // bytes of header the offset arithmetic below accounts for besides hsz
	int w, h;
	if (hsz == 12) {
		w = s.le16();
		h = s.le16();
	}
	else {
		w = (int)s.le32();
		h = (int)s.le32();
	}
	if (s.le16() != 1) return false;
	const int bpp = s.le16();
	uint32_t mr = 0, mg = 0, mb = 0, ma = 0;
	if (hsz != 12) {
		const int compress = (int)s.le32();
		if (compress == 1 || compress == 2) return false;
		s.skip(20);  // image size, resolution, colour counts
		if (hsz == 40 || hsz == 56) {
			if (hsz == 56) s.skip(16);
			if (bpp == 16 || bpp == 32) {
				if (compress == 0) {
					if (bpp == 32) {
						mr = 0xffu << 16;
						mg = 0xffu << 8;
						mb = 0xffu;
						ma = 0xffu << 24;
					}
					else {
						mr = 31u << 10;
						mg = 31u << 5;
						mb = 31u;
					}
				}
				else if (compress == 3) {
					mr = s.le32();
					mg = s.le32();
					mb = s.le32();
					consumed += 12;
					if (mr == mg && mg == mb) return false;
				}
				else
					return false;
			}
		}
		else {
			mr = s.le32();
			mg = s.le32();
			mb = s.le32();
			ma = s.le32();
			s.skip(52 + (hsz == 124 ? 16 : 0));  // colour space, end points, gamma; V5: intent and profile
		}
	}
	const bool bottomUp = h > 0;
	h = std::abs(h);
	if (!size_ok(w, h)) return false;
	img.w = w;
	img.h = h;
	img.rgb.assign((size_t)w * h * 3, 0);
	uint8_t* out = img.rgb.data();
	if (bpp < 16) {
		// The colour-table length follows from the pixel offset.  For the 12-byte header the reference subtracts 24
		// where 12 is meant (stb_image.h:5220), so it reads four entries fewer than the file holds; pixels that
		// index those are undefined there and black here.
		const int psize = hsz == 12 ? (offset - consumed - 24) / 3 : (offset - consumed - hsz) >> 2;
		if (psize <= 0 || psize > 256) return false;
		uint8_t pal[256][3];
		std::memset(pal, 0, sizeof pal);
		for (int i = 0; i < psize; ++i) {
			pal[i][2] = (uint8_t)s.u8();
			pal[i][1] = (uint8_t)s.u8();
			pal[i][0] = (uint8_t)s.u8();
			if (hsz != 12) s.u8();
		}
		s.skip(offset - consumed - hsz - psize * (hsz == 12 ? 3 : 4));
		int rowBytes;
		if (bpp == 1) rowBytes = (w + 7) >> 3;
		else if (bpp == 4) rowBytes = (w + 1) >> 1;
		else if (bpp == 8) rowBytes = w;
		else return false;
		const int pad = (-rowBytes) & 3;
		for (int y = 0; y < h; ++y) {
			int v = 0;
			for (int x = 0; x < w; ++x) {
				int idx;
				if (bpp == 8) idx = s.u8();
				else if (bpp == 4) {
					if ((x & 1) == 0) v = s.u8();
					idx = (x & 1) ? (v & 15) : (v >> 4);
				}
				else {
					if ((x & 7) == 0) v = s.u8();
					idx = (v >> (7 - (x & 7))) & 1;
				}
				std::memcpy(out, pal[idx], 3);
				out += 3;
			}
			s.skip(pad);
		}
	}
	else {
		s.skip(offset - consumed - hsz);
		if (bpp != 16 && bpp != 24 && bpp != 32) return false;
		const bool plain = bpp == 24 || (bpp == 32 && mb == 0xff && mg == 0xff00 && mr == 0x00ff0000 && ma == 0xff000000);
		int rs = 0, gs = 0, bs = 0, rc = 0, gc = 0, bc = 0;
		if (!plain) {
			if (!mr || !mg || !mb) return false;
			rs = top_bit(mr) - 7;
			rc = pop_count(mr);
			gs = top_bit(mg) - 7;
			gc = pop_count(mg);
			bs = top_bit(mb) - 7;
			bc = pop_count(mb);
			if (rc > 8 || gc > 8 || bc > 8) return false;  // the reference stops on an assertion here
		}
		const int pad = bpp == 24 ? (-3 * w) & 3 : bpp == 16 ? (-2 * w) & 3 : 0;
		for (int y = 0; y < h; ++y) {
			for (int x = 0; x < w; ++x, out += 3) {
				if (plain) {
					out[2] = (uint8_t)s.u8();
					out[1] = (uint8_t)s.u8();
					out[0] = (uint8_t)s.u8();
					if (bpp == 32) s.u8();
				}
				else {
					const uint32_t v = bpp == 16 ? (uint32_t)s.le16() : s.le32();
					out[0] = (uint8_t)widen_field(v & mr, rs, rc);
					out[1] = (uint8_t)widen_field(v & mg, gs, gc);
					out[2] = (uint8_t)widen_field(v & mb, bs, bc);
				}
			}
			s.skip(pad);
		}
	}
	if (bottomUp) {
		const size_t stride = (size_t)w * 3;
		std::vector<uint8_t> tmp(stride);
		for (int y = 0; y < h / 2; ++y) {
			uint8_t* a = &img.rgb[(size_t)y * stride];
			uint8_t* b = &img.rgb[(size_t)(h - 1 - y) * stride];
			std::memcpy(tmp.data(), a, stride);
			std::memcpy(a, b, stride);
			std::memcpy(b, tmp.data(), stride);
		}
	}
	return true;
}

// ---- GIF ---------------------------------------------------------------------------------------------------------
// First frame only, as the reference's single-image entry point returns it (stb_image.h:6646-6673 on top of
// 6394-6573): the canvas starts as transparent black, the frame's rectangle is painted over it, transparent pixels
// paint nothing, and with a non-zero background index every pixel the frame did not touch takes the background
// entry of the global table.

bool is_gif(const std::vector<uint8_t>& f) {
	return f.size() >= 6 && !std::memcmp(f.data(), "GIF8", 4) && (f[4] == '7' || f[4] == '9') && f[5] == 'a';
}

namespace {

struct GifTable {
	uint8_t rgba[256][4];
};

void gif_read_table(Bytes& s, GifTable& t, int entries, int transparent) {
	for (int i = 0; i < entries; ++i) {
		t.rgba[i][0] = (uint8_t)s.u8();
		t.rgba[i][1] = (uint8_t)s.u8();
		t.rgba[i][2] = (uint8_t)s.u8();
		t.rgba[i][3] = i == transparent ? 0 : 255;
	}
}

// Paints the code stream of one image descriptor into `canvas` (RGBA) and marks touched pixels.  Rows advance by
// `step` within [y0, y0 + rh); in interlaced pictures the four passes start at rows 0, 4, 2, 1 with steps 8, 8, 4, 2
// (stb_image.h:6267-6301 for the pixel cursor, 6303-6390 for the code stream).
struct GifPainter {
	std::vector<uint8_t>& canvas;
	std::vector<uint8_t>& touched;
	const GifTable& table;
	int cw, x0, y0, rw, rh;
	int x, y, step, passesLeft;
	bool done() const { return y >= y0 + rh; }
	void put(int colour) {
		if (done()) return;
		const size_t at = (size_t)y * cw + x;
		touched[at] = 1;
		const uint8_t* c = table.rgba[colour];
		if (c[3] > 128) std::memcpy(&canvas[at * 4], c, 4);
		if (++x >= x0 + rw) {
			x = x0;
			y += step;
			while (done() && passesLeft > 0) {
				step = 1 << passesLeft;
				y = y0 + (step >> 1);
				--passesLeft;
			}
		}
	}
};

bool gif_paint(Bytes& s, GifPainter& g) {
	struct Entry {
		int16_t prefix;
		uint8_t first, last;
	};
	const int minBits = s.u8();
	if (minBits > 12) return false;
	const int clear = 1 << minBits;
	std::vector<Entry> dict(8192, Entry{-1, 0, 0});
	for (int i = 0; i < clear; ++i) dict[i] = Entry{-1, (uint8_t)i, (uint8_t)i};
	std::vector<uint8_t> run(8192);
	int width = minBits + 1, mask = (1 << width) - 1;
	int next = clear + 2, prev = -1;
	bool seenClear = false;
	int32_t acc = 0;
	int have = 0, blockLeft = 0;
	for (;;) {
		if (have < width) {
			if (blockLeft == 0) {
				blockLeft = s.u8();
				if (blockLeft == 0) return true;  // block terminator: the picture ends where the data does
			}
			--blockLeft;
			acc |= (int32_t)s.u8() << have;
			have += 8;
			continue;
		}
		const int code = acc & mask;
		acc >>= width;
		have -= width;
		if (code == clear) {
			width = minBits + 1;
			mask = (1 << width) - 1;
			next = clear + 2;
			prev = -1;
			seenClear = true;
		}
		else if (code == clear + 1) {
			s.skip(blockLeft);
			while ((blockLeft = s.u8()) > 0) s.skip(blockLeft);
			return true;
		}
		else if (code <= next) {
			if (!seenClear) return false;
			if (prev >= 0) {
				if (next + 1 > 8192) return false;
				Entry& e = dict[next++];
				e.prefix = (int16_t)prev;
				e.first = dict[prev].first;
				e.last = code == next ? e.first : dict[code].first;
			}
			else if (code == next)
				return false;
			int len = 0;
			for (int c = code; c >= 0 && len < 8192; c = dict[c].prefix) run[len++] = dict[c].last;
			while (len > 0) g.put(run[--len]);
			if ((next & mask) == 0 && next <= 0x0FFF) {
				++width;
				mask = (1 << width) - 1;
			}
			prev = code;
		}
		else
			return false;
	}
}

}  // namespace

bool load_gif(const std::vector<uint8_t>& f, Image& img) {
	if (!is_gif(f)) return false;
	Bytes s(f);
	s.skip(6);
	const int cw = s.le16(), chh = s.le16();
	const int flags = s.u8(), bgIndex = s.u8();
	s.u8();  // aspect ratio
	GifTable global, local;
	std::memset(&global, 0, sizeof global);
	std::memset(&local, 0, sizeof local);
	if (flags & 0x80) gif_read_table(s, global, 2 << (flags & 7), -1);
	if (!size_ok(cw, chh)) return false;
	std::vector<uint8_t> canvas((size_t)cw * chh * 4, 0), touched((size_t)cw * chh, 0);
	int control = 0, transparent = -1;
	for (;;) {
		const int tag = s.u8();
		if (tag == 0x2C) {
			const int x = s.le16(), y = s.le16(), rw = s.le16(), rh = s.le16();
			if (x + rw > cw || y + rh > chh) return false;
			const int lflags = s.u8();
			const GifTable* table;
			if (lflags & 0x80) {
				gif_read_table(s, local, 2 << (lflags & 7), (control & 1) ? transparent : -1);
				table = &local;
			}
			else if (flags & 0x80)
				table = &global;
			else
				return false;
			GifPainter g{canvas, touched, *table, cw, x, y, rw, rh, x, y, (lflags & 0x40) ? 8 : 1, (lflags & 0x40) ? 3 : 0};
			if (rw == 0) g.y = y + rh;  // an empty rectangle paints nothing
			if (!gif_paint(s, g)) return false;
			img.w = cw;
			img.h = chh;
			img.rgb.resize((size_t)cw * chh * 3);
			// Untouched pixels: the reference copies the table entry as stored, blue first, into its RGBA canvas
			// (stb_image.h:6516-6523), so red and blue arrive swapped; the same is done here.
			const uint8_t bg[3] = {global.rgba[bgIndex][2], global.rgba[bgIndex][1], global.rgba[bgIndex][0]};
			for (size_t i = 0; i < touched.size(); ++i) {
				const uint8_t* src = (bgIndex > 0 && !touched[i]) ? bg : &canvas[i * 4];
				std::memcpy(&img.rgb[i * 3], src, 3);
			}
			return true;
		}
		else if (tag == 0x21) {
			const int label = s.u8();
			int len;
			if (label == 0xF9) {  // graphic control: which colour index is transparent
				len = s.u8();
				if (len != 4) {
					s.skip(len);
					continue;
				}
				control = s.u8();
				s.le16();  // frame delay
				if (transparent >= 0) global.rgba[transparent][3] = 255;
				if (control & 1) {
					transparent = s.u8();
					global.rgba[transparent][3] = 0;
				}
				else {
					s.u8();
					transparent = -1;
				}
			}
			while ((len = s.u8()) != 0) s.skip(len);
		}
		else
			return false;  // trailer before any picture, or an unknown block
	}
}

// ---- PSD ---------------------------------------------------------------------------------------------------------
// The flattened picture of an RGB Photoshop file, 8 or 16 bits a channel, raw or PackBits rows
// (stb_image.h:5756-5959, 5717-5754).

bool is_psd(const std::vector<uint8_t>& f) { return f.size() >= 4 && !std::memcmp(f.data(), "8BPS", 4); }

bool load_psd(const std::vector<uint8_t>& f, Image& img) {
	if (!is_psd(f)) return false;
	Bytes s(f);
	s.skip(4);
	if (s.be16() != 1) return false;
	s.skip(6);
	const int channels = s.be16();
	if (channels < 0 || channels > 16) return false;
	const int h = (int)s.be32(), w = (int)s.be32();
	const int depth = s.be16();
	if (depth != 8 && depth != 16) return false;
	if (s.be16() != 3) return false;  // colour mode RGB
	s.skip(s.be32());                 // mode data
	s.skip(s.be32());                 // image resources
	s.skip(s.be32());                 // layers and masks
	const int compression = s.be16();
	if (compression > 1) return false;
	if (!size_ok(w, h)) return false;
	const size_t count = (size_t)w * h;
	std::vector<uint8_t> px(count * 4);
	if (compression) s.skip((long long)h * channels * 2);  // the table of packed row lengths
	for (int c = 0; c < 4; ++c) {
		uint8_t* p = px.data() + c;
		if (c >= channels) {
			for (size_t i = 0; i < count; ++i) p[i * 4] = c == 3 ? 255 : 0;
		}
		else if (compression) {
			size_t done = 0;
			while (done < count) {
				int len = s.u8();
				if (len == 128) continue;
				if (len < 128) {
					++len;
					if ((size_t)len > count - done) return false;
					for (int k = 0; k < len; ++k) p[(done++) * 4] = (uint8_t)s.u8();
				}
				else {
					len = 257 - len;
					if ((size_t)len > count - done) return false;
					const uint8_t v = (uint8_t)s.u8();
					for (int k = 0; k < len; ++k) p[(done++) * 4] = v;
				}
			}
		}
		else if (depth == 16) {
			for (size_t i = 0; i < count; ++i) p[i * 4] = (uint8_t)(s.be16() >> 8);
		}
		else {
			for (size_t i = 0; i < count; ++i) p[i * 4] = (uint8_t)s.u8();
		}
	}
	img.w = w;
	img.h = h;
	img.rgb.resize(count * 3);
	for (size_t i = 0; i < count; ++i) {
		uint8_t* q = &px[i * 4];
		if (channels >= 4 && q[3] != 0 && q[3] != 255) {
			// Photoshop composites partly transparent pixels over white; take the white out again
			const float a = q[3] / 255.0f;
			const float ra = 1.0f / a;
			const float inv = 255.0f * (1 - ra);
			for (int k = 0; k < 3; ++k) q[k] = (uint8_t)(int)(q[k] * ra + inv);
		}
		std::memcpy(&img.rgb[i * 3], q, 3);
	}
	return true;
}

// ---- PIC ---------------------------------------------------------------------------------------------------------
// Softimage PIC: up to ten channel packets a row, raw, run-length or mixed (stb_image.h:6019-6164).

bool is_pic(const std::vector<uint8_t>& f) {
	static const uint8_t magic[4] = {0x53, 0x80, 0xF6, 0x34};
	return f.size() >= 92 && !std::memcmp(f.data(), magic, 4) && !std::memcmp(f.data() + 88, "PICT", 4);
}

namespace {

bool pic_value(Bytes& s, int channelMask, uint8_t* dst) {
	for (int i = 0, bit = 0x80; i < 4; ++i, bit >>= 1)
		if (channelMask & bit) {
			if (s.eof()) return false;
			dst[i] = (uint8_t)s.u8();
		}
	return true;
}

void pic_copy(int channelMask, uint8_t* dst, const uint8_t* src) {
	for (int i = 0, bit = 0x80; i < 4; ++i, bit >>= 1)
		if (channelMask & bit) dst[i] = src[i];
}

}  // namespace

bool load_pic(const std::vector<uint8_t>& f, Image& img) {
	if (!is_pic(f)) return false;
	Bytes s(f);
	s.skip(92);
	const int w = s.be16(), h = s.be16();
	if (s.eof() || !size_ok(w, h)) return false;
	s.skip(8);  // pixel ratio, field type, padding
	struct Packet {
		int size, type, channels;
	} packets[10];
	int nPackets = 0, chained;
	do {
		if (nPackets == 10) return false;
		Packet& p = packets[nPackets++];
		chained = s.u8();
		p.size = s.u8();
		p.type = s.u8();
		p.channels = s.u8();
		if (s.eof() || p.size != 8) return false;
	} while (chained);
	std::vector<uint8_t> px((size_t)w * h * 4, 0xff);
	for (int y = 0; y < h; ++y)
		for (int k = 0; k < nPackets; ++k) {
			const Packet& p = packets[k];
			uint8_t* dst = &px[(size_t)y * w * 4];
			if (p.type == 0) {
				for (int x = 0; x < w; ++x, dst += 4)
					if (!pic_value(s, p.channels, dst)) return false;
			}
			else if (p.type == 1) {
				int left = w;
				while (left > 0) {
					int count = s.u8();
					if (s.eof()) return false;
					if (count > left) count = left;
					uint8_t v[4];
					if (!pic_value(s, p.channels, v)) return false;
					for (int i = 0; i < count; ++i, dst += 4) pic_copy(p.channels, dst, v);
					left -= count;
				}
			}
			else if (p.type == 2) {
				int left = w;
				while (left > 0) {
					int count = s.u8();
					if (s.eof()) return false;
					if (count >= 128) {
						count = count == 128 ? s.be16() : count - 127;
						if (count > left) return false;
						uint8_t v[4];
						if (!pic_value(s, p.channels, v)) return false;
						for (int i = 0; i < count; ++i, dst += 4) pic_copy(p.channels, dst, v);
					}
					else {
						++count;
						if (count > left) return false;
						for (int i = 0; i < count; ++i, dst += 4)
							if (!pic_value(s, p.channels, dst)) return false;
					}
					left -= count;
				}
			}
			else
				return false;
		}
	img.w = w;
	img.h = h;
	img.rgb.resize((size_t)w * h * 3);
	for (size_t i = 0; i < (size_t)w * h; ++i) std::memcpy(&img.rgb[i * 3], &px[i * 4], 3);
	return true;
}

// ---- Radiance HDR --------------------------------------------------------------------------------------------------
// RGBE pixels, flat or with the per-channel run-length rows, to float and from there to bytes with the reference's
// display curve: (v ^ (1/2.2)) * 255 + 0.5, clamped and truncated (stb_image.h:6752-6879 and 1712-1736).

bool is_hdr(const std::vector<uint8_t>& f) {
	return (f.size() >= 11 && !std::memcmp(f.data(), "#?RADIANCE\n", 11)) || (f.size() >= 7 && !std::memcmp(f.data(), "#?RGBE\n", 7));
}

namespace {

std::string hdr_line(Bytes& s) {
	std::string line;
	int c = s.u8();
	while (!s.eof() && c != '\n') {
		line.push_back((char)c);
		if (line.size() == 1023) {
			while (!s.eof() && s.u8() != '\n') {
			}
			break;
		}
		c = s.u8();
	}
	return line;
}

void hdr_pixel(float* out, const uint8_t* rgbe) {
	if (rgbe[3] != 0) {
		const float scale = (float)std::ldexp(1.0f, rgbe[3] - (128 + 8));
		for (int k = 0; k < 3; ++k) out[k] = rgbe[k] * scale;
	}
	else
		out[0] = out[1] = out[2] = 0;
}

}  // namespace

bool load_hdr(const std::vector<uint8_t>& f, Image& img) {
	if (!is_hdr(f)) return false;
	Bytes s(f);
	const std::string id = hdr_line(s);
	if (id != "#?RADIANCE" && id != "#?RGBE") return false;
	bool rgbeFormat = false;
	for (;;) {
		const std::string line = hdr_line(s);
		if (line.empty()) break;
		if (line == "FORMAT=32-bit_rle_rgbe") rgbeFormat = true;
	}
	if (!rgbeFormat) return false;
	const std::string dims = hdr_line(s);
	if (dims.compare(0, 3, "-Y ") != 0) return false;
	char* end = nullptr;
	const int h = (int)std::strtol(dims.c_str() + 3, &end, 10);
	while (*end == ' ') ++end;
	if (std::strncmp(end, "+X ", 3) != 0) return false;
	const int w = (int)std::strtol(end + 3, nullptr, 10);
	if (!size_ok(w, h)) return false;
	std::vector<float> lin((size_t)w * h * 3);
	size_t flatFrom = 0;       // first pixel to read as a flat RGBE quadruple
	bool flat = w < 8 || w >= 32768;
	if (!flat) {
		std::vector<uint8_t> row((size_t)w * 4);
		for (int y = 0; y < h && !flat; ++y) {
			const int c1 = s.u8(), c2 = s.u8();
			int len = s.u8();
			if (c1 != 2 || c2 != 2 || (len & 0x80)) {
				// Not a run-length row: the three bytes and the next are pixel 0 and the rest of the file is flat,
				// restarting at the top of the picture whatever row this is (stb_image.h:6825-6838)
				const uint8_t rgbe[4] = {(uint8_t)c1, (uint8_t)c2, (uint8_t)len, (uint8_t)s.u8()};
				hdr_pixel(lin.data(), rgbe);
				flat = true;
				flatFrom = 1;
				break;
			}
			len = (len << 8) | s.u8();
			if (len != w) return false;
			for (int k = 0; k < 4; ++k) {
				int x = 0;
				while (x < w) {
					int count = s.u8();
					if (count > 128) {
						const uint8_t v = (uint8_t)s.u8();
						count -= 128;
						if (count > w - x) return false;
						for (int z = 0; z < count; ++z) row[(size_t)(x++) * 4 + k] = v;
					}
					else {
						if (count > w - x) return false;
						for (int z = 0; z < count; ++z) row[(size_t)(x++) * 4 + k] = (uint8_t)s.u8();
					}
					if (count == 0 && s.eof()) return false;  // a zero count at the end of the file would never advance
				}
			}
			for (int x = 0; x < w; ++x) hdr_pixel(&lin[((size_t)y * w + x) * 3], &row[(size_t)x * 4]);
		}
	}
	if (flat) {
		uint8_t rgbe[4] = {0, 0, 0, 0};
		for (size_t i = flatFrom; i < (size_t)w * h; ++i) {
			if (s.n - s.pos >= 4) {  // a short read leaves the previous quadruple in place, as the reference's does
				std::memcpy(rgbe, s.p + s.pos, 4);
				s.pos += 4;
			}
			else
				s.pos = s.n;
			hdr_pixel(&lin[i * 3], rgbe);
		}
	}
	img.w = w;
	img.h = h;
	img.rgb.resize(lin.size());
	const float gamma = 1.0f / 2.2f, scale = 1.0f;
	for (size_t i = 0; i < lin.size(); ++i) {
		float z = (float)std::pow(lin[i] * scale, gamma) * 255 + 0.5f;
		if (z < 0) z = 0;
		if (z > 255) z = 255;
		img.rgb[i] = (uint8_t)(int)z;
	}
	return true;
}

}  // namespace rtxh
