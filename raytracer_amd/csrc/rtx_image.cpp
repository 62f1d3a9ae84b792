// Image file I/O for the host side: 8-bit RGB texture loading (binary PPM/PGM, uncompressed or RLE TGA, PNG,
// JPEG via rtx_jpeg.cpp, BMP/GIF/PSD/PIC/HDR via rtx_image_formats.cpp) with the reference's vertical flip
// (src/Texture.cpp:6-7: stbi_set_flip_vertically_on_load(1), 3 channels forced), and the PNG
// writer behind out_%03d.png (src/RayTracer.cpp:263-269: 3 channels, stride 3*width).
// Self-contained: no image library is linked.

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rtx_host_internal.h"
#include "rtx_image.h"

namespace rtxh {

namespace {

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
	FILE* f = std::fopen(path.c_str(), "rb");
	if (!f) return false;
	std::fseek(f, 0, SEEK_END);
	long n = std::ftell(f);
	std::fseek(f, 0, SEEK_SET);
	if (n < 0) {
		std::fclose(f);
		return false;
	}
	out.resize((size_t)n);
	size_t got = n ? std::fread(out.data(), 1, (size_t)n, f) : 0;
	std::fclose(f);
	return got == (size_t)n;
}

// ---- CRC32 / Adler32 -------------------------------------------------------------------------
// CRC-32 eight bytes at a time (tables 1..7 are table 0 advanced by one to seven zero bytes): the image writer runs beside
// the renderer and has a frame's six megabytes to get through in less than the frame takes to render.
struct CrcTables {
	uint32_t t[8][256];
	CrcTables() {
		for (uint32_t n = 0; n < 256; ++n) {
			uint32_t c = n;
			for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
			t[0][n] = c;
		}
		for (int k = 1; k < 8; ++k)
			for (uint32_t n = 0; n < 256; ++n) t[k][n] = (t[k - 1][n] >> 8) ^ t[0][t[k - 1][n] & 0xFF];
	}
};
const CrcTables& crc_tables() {
	static const CrcTables tables;  // (initialised once, whichever thread comes first)
	return tables;
}
uint32_t crc32(uint32_t crc, const uint8_t* p, size_t n) {
	const CrcTables& T = crc_tables();
	crc = ~crc;
	while (n >= 8) {
		const uint32_t lo = crc ^ ((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24);
		const uint32_t hi = (uint32_t)p[4] | (uint32_t)p[5] << 8 | (uint32_t)p[6] << 16 | (uint32_t)p[7] << 24;
		crc = T.t[7][lo & 0xFF] ^ T.t[6][(lo >> 8) & 0xFF] ^ T.t[5][(lo >> 16) & 0xFF] ^ T.t[4][lo >> 24] ^
			  T.t[3][hi & 0xFF] ^ T.t[2][(hi >> 8) & 0xFF] ^ T.t[1][(hi >> 16) & 0xFF] ^ T.t[0][hi >> 24];
		p += 8;
		n -= 8;
	}
	for (size_t i = 0; i < n; ++i) crc = T.t[0][(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
	return ~crc;
}
// Adler-32 over a stream handed in pieces; the sums are reduced every 5552 bytes, the most that cannot overflow 32 bits.
struct Adler32 {
	uint32_t a = 1, b = 0;
	void add(const uint8_t* p, size_t n) {
		while (n) {
			const size_t k = n < 5552 ? n : 5552;
			size_t i = 0;
			for (; i + 8 <= k; i += 8) {
				a += p[i]; b += a; a += p[i + 1]; b += a; a += p[i + 2]; b += a; a += p[i + 3]; b += a;
				a += p[i + 4]; b += a; a += p[i + 5]; b += a; a += p[i + 6]; b += a; a += p[i + 7]; b += a;
			}
			for (; i < k; ++i) {
				a += p[i];
				b += a;
			}
			a %= 65521;
			b %= 65521;
			p += k;
			n -= k;
		}
	}
	uint32_t value() const { return (b << 16) | a; }
};

// ---- inflate (RFC 1951), enough for PNG ----------------------------------------------------
struct BitReader {
	const uint8_t* p;
	size_t n, pos = 0;
	uint32_t buf = 0;
	int cnt = 0;
	bool need(int k) {
		while (cnt < k) {
			if (pos >= n) return false;
			buf |= (uint32_t)p[pos++] << cnt;
			cnt += 8;
		}
		return true;
	}
	bool bits(int k, uint32_t& v) {
		if (k == 0) {
			v = 0;
			return true;
		}
		if (!need(k)) return false;
		v = buf & ((1u << k) - 1);
		buf >>= k;
		cnt -= k;
		return true;
	}
};

struct Huff {
	uint16_t count[16];
	uint16_t symbol[320];
	bool build(const uint8_t* len, int n) {
		std::memset(count, 0, sizeof(count));
		for (int i = 0; i < n; ++i) count[len[i]]++;
		uint16_t offs[16];
		offs[1] = 0;
		for (int i = 1; i < 15; ++i) offs[i + 1] = offs[i] + count[i];
		for (int i = 0; i < n; ++i)
			if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
		return true;
	}
	int decode(BitReader& br) const {
		int code = 0, first = 0, index = 0;
		for (int len = 1; len <= 15; ++len) {
			uint32_t b;
			if (!br.bits(1, b)) return -1;
			code |= (int)b;
			int c = count[len];
			if (code - c < first) return symbol[index + (code - first)];
			index += c;
			first += c;
			first <<= 1;
			code <<= 1;
		}
		return -1;
	}
};

bool inflate_zlib(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {
	if (n < 6) return false;
	BitReader br{src + 2, n - 2};
	static const uint16_t lbase[] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
	static const uint16_t lext[] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
	static const uint16_t dbase[] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
	static const uint16_t dext[] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
	uint32_t last = 0;
	while (!last) {
		uint32_t type;
		if (!br.bits(1, last) || !br.bits(2, type)) return false;
		if (type == 0) {
			br.buf = 0;
			br.cnt = 0;
			if (br.pos + 4 > br.n) return false;
			uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8);
			br.pos += 4;
			if (br.pos + len > br.n) return false;
			out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
			br.pos += len;
			continue;
		}
		Huff lit, dist;
		uint8_t lens[320];
		if (type == 1) {
			int i = 0;
			for (; i < 144; ++i) lens[i] = 8;
			for (; i < 256; ++i) lens[i] = 9;
			for (; i < 280; ++i) lens[i] = 7;
			for (; i < 288; ++i) lens[i] = 8;
			lit.build(lens, 288);
			for (i = 0; i < 30; ++i) lens[i] = 5;
			dist.build(lens, 30);
		}
		else if (type == 2) {
			uint32_t hlit, hdist, hclen;
			if (!br.bits(5, hlit) || !br.bits(5, hdist) || !br.bits(4, hclen)) return false;
			hlit += 257;
			hdist += 1;
			hclen += 4;
			static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
			uint8_t cl[19] = {0};
			for (uint32_t i = 0; i < hclen; ++i) {
				uint32_t v;
				if (!br.bits(3, v)) return false;
				cl[order[i]] = (uint8_t)v;
			}
			Huff clh;
			clh.build(cl, 19);
			uint32_t idx = 0;
			while (idx < hlit + hdist) {
				int sym = clh.decode(br);
				if (sym < 0) return false;
				if (sym < 16) lens[idx++] = (uint8_t)sym;
				else {
					uint32_t rep, prev = 0;
					if (sym == 16) {
						if (idx == 0) return false;
						prev = lens[idx - 1];
						if (!br.bits(2, rep)) return false;
						rep += 3;
					}
					else if (sym == 17) {
						if (!br.bits(3, rep)) return false;
						rep += 3;
					}
					else {
						if (!br.bits(7, rep)) return false;
						rep += 11;
					}
					if (idx + rep > hlit + hdist) return false;
					while (rep--) lens[idx++] = (uint8_t)prev;
				}
			}
			lit.build(lens, (int)hlit);
			dist.build(lens + hlit, (int)hdist);
		}
		else
			return false;
		for (;;) {
			int sym = lit.decode(br);
			if (sym < 0) return false;
			if (sym < 256) out.push_back((uint8_t)sym);
			else if (sym == 256) break;
			else {
				sym -= 257;
				if (sym >= 29) return false;
				uint32_t eb;
				if (!br.bits(lext[sym], eb)) return false;
				uint32_t len = lbase[sym] + eb;
				int ds = dist.decode(br);
				if (ds < 0 || ds >= 30) return false;
				if (!br.bits(dext[ds], eb)) return false;
				uint32_t d = dbase[ds] + eb;
				if (d > out.size()) return false;
				size_t from = out.size() - d;
				for (uint32_t i = 0; i < len; ++i) out.push_back(out[from + i]);
			}
		}
	}
	return true;
}

bool load_png(const std::vector<uint8_t>& f, Image& img) {
	static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	if (f.size() < 8 || std::memcmp(f.data(), sig, 8)) return false;
	size_t pos = 8;
	uint32_t w = 0, h = 0;
	int depth = 0, ctype = 0, interlace = 0;
	std::vector<uint8_t> idat, plte;
	while (pos + 8 <= f.size()) {
		uint32_t len = (f[pos] << 24) | (f[pos + 1] << 16) | (f[pos + 2] << 8) | f[pos + 3];
		const char* type = (const char*)&f[pos + 4];
		if (pos + 12 + len > f.size()) return false;
		const uint8_t* d = &f[pos + 8];
		if (!std::memcmp(type, "IHDR", 4)) {
			w = (d[0] << 24) | (d[1] << 16) | (d[2] << 8) | d[3];
			h = (d[4] << 24) | (d[5] << 16) | (d[6] << 8) | d[7];
			depth = d[8];
			ctype = d[9];
			interlace = d[12];
		}
		else if (!std::memcmp(type, "PLTE", 4)) plte.assign(d, d + len);
		else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
		else if (!std::memcmp(type, "IEND", 4)) break;
		pos += 12 + len;
	}
	if (!w || !h || interlace > 1 || (uint64_t)w * h > (1ull << 28)) return false;
	const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
	if (!ch) return false;
	if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) return false;
	if (depth < 8 && ctype != 0 && ctype != 3) return false;
	if (depth == 16 && ctype == 3) return false;
	std::vector<uint8_t> raw;
	if (!inflate_zlib(idat.data(), idat.size(), raw)) return false;
	img.w = (int)w;
	img.h = (int)h;
	img.rgb.assign((size_t)w * h * 3, 0);
	// sample -> byte: 16-bit samples keep their high byte, grey samples below 8 bits are spread over 0..255,
	// palette indices stay as they are (what stb_image does for an 8-bit load)
	const int greyScale = (ctype == 0 && depth < 8) ? (depth == 1 ? 255 : depth == 2 ? 85 : 17) : 1;
	auto put = [&](uint32_t x, uint32_t y, const uint8_t* s8) {
		uint8_t* dst = &img.rgb[((size_t)y * w + x) * 3];
		if (ctype == 3) {
			const size_t k = (size_t)s8[0] * 3;
			if (k + 2 < plte.size()) {
				dst[0] = plte[k];
				dst[1] = plte[k + 1];
				dst[2] = plte[k + 2];
			}
		}
		else if (ch <= 2) dst[0] = dst[1] = dst[2] = s8[0];
		else {
			dst[0] = s8[0];
			dst[1] = s8[1];
			dst[2] = s8[2];
		}
	};
	// the whole image, or the seven Adam7 passes: (x0, y0, dx, dy)
	static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
	size_t at = 0;
	const size_t bpp = (size_t)std::max(1, ch * depth / 8);  // filter distance in bytes
	for (int pass = 0; pass < (interlace ? 7 : 1); ++pass) {
		const int x0 = interlace ? adam7[pass][0] : 0, y0 = interlace ? adam7[pass][1] : 0;
		const int dx = interlace ? adam7[pass][2] : 1, dy = interlace ? adam7[pass][3] : 1;
		if ((uint32_t)x0 >= w || (uint32_t)y0 >= h) continue;
		const uint32_t pw = (w - x0 + dx - 1) / dx, ph = (h - y0 + dy - 1) / dy;
		const size_t rowBytes = ((size_t)pw * ch * depth + 7) / 8;
		if (raw.size() < at + (size_t)ph * (rowBytes + 1)) return false;
		std::vector<uint8_t> cur(rowBytes), prev(rowBytes, 0);
		for (uint32_t py = 0; py < ph; ++py) {
			const uint8_t* row = &raw[at];
			at += rowBytes + 1;
			const int ft = row[0];
			for (size_t i = 0; i < rowBytes; ++i) {
				const int a = i >= bpp ? cur[i - bpp] : 0, b2 = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
				const int x = row[1 + i];
				int v;
				switch (ft) {
				case 0: v = x; break;
				case 1: v = x + a; break;
				case 2: v = x + b2; break;
				case 3: v = x + ((a + b2) >> 1); break;
				case 4: {
					const int pr = a + b2 - c, pa = std::abs(pr - a), pb = std::abs(pr - b2), pc = std::abs(pr - c);
					v = x + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b2 : c));
					break;
				}
				default: return false;
				}
				cur[i] = (uint8_t)v;
			}
			for (uint32_t px = 0; px < pw; ++px) {
				uint8_t s8[4] = {0, 0, 0, 0};
				if (depth == 8)
					for (int k = 0; k < ch; ++k) s8[k] = cur[(size_t)px * ch + k];
				else if (depth == 16)
					for (int k = 0; k < ch; ++k) s8[k] = cur[((size_t)px * ch + k) * 2];
				else {
					const size_t bit = (size_t)px * depth;
					const int v = (cur[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1 << depth) - 1);
					s8[0] = (uint8_t)(v * greyScale);
				}
				put((uint32_t)x0 + px * dx, (uint32_t)y0 + py * dy, s8);
			}
			prev.swap(cur);
		}
	}
	return true;
}

bool load_pnm(const std::vector<uint8_t>& f, Image& img) {
	if (f.size() < 7 || f[0] != 'P' || (f[1] != '6' && f[1] != '5')) return false;
	const int ch = f[1] == '6' ? 3 : 1;
	size_t pos = 2;
	int vals[3], got = 0;
	while (got < 3 && pos < f.size()) {
		while (pos < f.size() && (f[pos] == ' ' || f[pos] == '\n' || f[pos] == '\r' || f[pos] == '\t')) ++pos;
		if (pos < f.size() && f[pos] == '#') {
			while (pos < f.size() && f[pos] != '\n') ++pos;
			continue;
		}
		int v = 0;
		bool any = false;
		while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9') {
			v = v * 10 + (f[pos++] - '0');
			any = true;
		}
		if (!any) return false;
		vals[got++] = v;
	}
	if (got < 3 || vals[2] != 255) return false;
	++pos;  // the single whitespace byte after maxval
	const size_t n = (size_t)vals[0] * vals[1];
	if (pos + n * ch > f.size()) return false;
	img.w = vals[0];
	img.h = vals[1];
	img.rgb.resize(n * 3);
	for (size_t i = 0; i < n; ++i)
		for (int c = 0; c < 3; ++c) img.rgb[i * 3 + c] = f[pos + i * ch + (ch == 3 ? c : 0)];
	return true;
}

bool load_tga(const std::vector<uint8_t>& f, Image& img) {
	if (f.size() < 18) return false;
	const int idlen = f[0], cmap = f[1], type = f[2];
	const int w = f[12] | (f[13] << 8), h = f[14] | (f[15] << 8), bpp = f[16], desc = f[17];
	if (cmap != 0 || (type != 2 && type != 10 && type != 3 && type != 11) || w <= 0 || h <= 0) return false;
	const int ch = bpp / 8;
	if (ch != 1 && ch != 3 && ch != 4) return false;
	size_t pos = 18 + idlen;
	std::vector<uint8_t> px((size_t)w * h * ch);
	if (type == 2 || type == 3) {
		if (pos + px.size() > f.size()) return false;
		std::memcpy(px.data(), &f[pos], px.size());
	}
	else {
		size_t o = 0;
		while (o < px.size()) {
			if (pos >= f.size()) return false;
			int hdr = f[pos++];
			int cnt = (hdr & 0x7F) + 1;
			if (hdr & 0x80) {
				if (pos + ch > f.size()) return false;
				for (int i = 0; i < cnt && o < px.size(); ++i, o += ch) std::memcpy(&px[o], &f[pos], ch);
				pos += ch;
			}
			else {
				size_t bytes = (size_t)cnt * ch;
				if (pos + bytes > f.size() || o + bytes > px.size()) return false;
				std::memcpy(&px[o], &f[pos], bytes);
				pos += bytes;
				o += bytes;
			}
		}
	}
	const bool topDown = (desc & 0x20) != 0;
	img.w = w;
	img.h = h;
	img.rgb.resize((size_t)w * h * 3);
	for (int y = 0; y < h; ++y) {
		const uint8_t* row = &px[(size_t)(topDown ? y : h - 1 - y) * w * ch];
		for (int x = 0; x < w; ++x) {
			uint8_t* d = &img.rgb[((size_t)y * w + x) * 3];
			const uint8_t* p = row + (size_t)x * ch;
			if (ch == 1) d[0] = d[1] = d[2] = p[0];
			else {
				d[0] = p[2];
				d[1] = p[1];
				d[2] = p[0];
			}
		}
	}
	return true;
}

}  // namespace

// Decodes to top-down RGB, then flips vertically: after this call row 0 is the bottom row of
// the picture, which is what Texture::data holds in the reference.
// The container is chosen by signature in the reference's order (stb_image.h:1019-1063: JPEG, PNG, BMP, GIF, PSD,
// PIC, PNM, HDR and, having no signature, TGA last); a file that carries a signature and fails to decode is not
// offered to the later decoders.
bool load_texture_file(const std::string& path, Image& img) {
	std::vector<uint8_t> f;
	if (!read_file(path, f)) return false;
	static const uint8_t pngSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	bool ok;
	if (f.size() >= 2 && f[0] == 0xFF && f[1] == 0xD8) ok = load_jpeg(f, img);
	else if (f.size() >= 8 && !std::memcmp(f.data(), pngSig, 8)) ok = load_png(f, img);
	else if (is_bmp(f)) ok = load_bmp(f, img);
	else if (is_gif(f)) ok = load_gif(f, img);
	else if (is_psd(f)) ok = load_psd(f, img);
	else if (is_pic(f)) ok = load_pic(f, img);
	else if (f.size() >= 2 && f[0] == 'P' && (f[1] == '5' || f[1] == '6')) ok = load_pnm(f, img);
	else if (is_hdr(f)) ok = load_hdr(f, img);
	else ok = load_tga(f, img);
	if (!ok) return false;
	const size_t stride = (size_t)img.w * 3;
	std::vector<uint8_t> tmp(stride);
	for (int y = 0; y < img.h / 2; ++y) {
		uint8_t* a = &img.rgb[(size_t)y * stride];
		uint8_t* b = &img.rgb[(size_t)(img.h - 1 - y) * stride];
		std::memcpy(tmp.data(), a, stride);
		std::memcpy(a, b, stride);
		std::memcpy(b, tmp.data(), stride);
	}
	return true;
}

// 8-bit RGB PNG.  Scanlines are stored in zlib "stored" blocks: the frame loop writes one image
// per frame (src/RayTracer.cpp:269) and must not wait on a compressor.
bool write_png(const std::string& path, int w, int h, const uint8_t* rgb) {
	// The file is laid out in one buffer of its final size and every byte of the image is touched three times (copy, Adler-32,
	// CRC-32): 1920x1080 in about 10 ms on one host thread, where the frame takes 22 ms to render.
	const size_t stride = (size_t)w * 3, line = stride + 1, rawSize = (size_t)h * line;
	const size_t nBlocks = rawSize ? (rawSize + 65534) / 65535 : 1;
	const size_t zSize = 2 + nBlocks * 5 + rawSize + 4;
	std::vector<uint8_t> png(8 + (12 + 13) + (12 + zSize) + 12);
	uint8_t* q = png.data();
	auto be32 = [&](uint32_t x) {
		*q++ = (uint8_t)(x >> 24); *q++ = (uint8_t)(x >> 16); *q++ = (uint8_t)(x >> 8); *q++ = (uint8_t)x;
	};
	static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	std::memcpy(q, sig, 8);
	q += 8;
	be32(13);
	uint8_t* from = q;
	std::memcpy(q, "IHDR", 4);
	q += 4;
	be32((uint32_t)w);
	be32((uint32_t)h);
	*q++ = 8; *q++ = 2; *q++ = 0; *q++ = 0; *q++ = 0;
	be32(crc32(0, from, (size_t)(q - from)));
	be32((uint32_t)zSize);
	from = q;
	std::memcpy(q, "IDAT", 4);
	q += 4;
	*q++ = 0x78;
	*q++ = 0x01;
	// the scanlines (a zero filter byte, then the row) cut into stored blocks of at most 65,535 bytes
	Adler32 sum;
	static const uint8_t zero = 0;
	size_t pos = 0;  // in the stream of scanlines
	for (size_t b = 0; b < nBlocks; ++b) {
		size_t n = rawSize - pos < 65535 ? rawSize - pos : 65535;
		*q++ = pos + n == rawSize ? 1 : 0;
		*q++ = (uint8_t)(n & 0xFF);
		*q++ = (uint8_t)(n >> 8);
		*q++ = (uint8_t)(~n & 0xFF);
		*q++ = (uint8_t)((~n >> 8) & 0xFF);
		while (n) {
			const size_t row = pos / line, off = pos % line;
			if (off == 0) {
				*q++ = 0;
				sum.add(&zero, 1);
				++pos;
				--n;
				continue;
			}
			const size_t span = n < line - off ? n : line - off;
			const uint8_t* src = rgb + row * stride + (off - 1);
			std::memcpy(q, src, span);
			sum.add(src, span);
			q += span;
			pos += span;
			n -= span;
		}
	}
	be32(sum.value());
	be32(crc32(0, from, (size_t)(q - from)));
	be32(0);
	from = q;
	std::memcpy(q, "IEND", 4);
	q += 4;
	be32(crc32(0, from, 4));
	if ((size_t)(q - png.data()) != png.size()) return false;
	FILE* f = std::fopen(path.c_str(), "wb");
	if (!f) return false;
	bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
	if (std::fclose(f) != 0) ok = false;
	return ok;
}

}  // namespace rtxh
