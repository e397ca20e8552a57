/*
 * nhw-enc -- drop-in command line encoder on top of libnhwhip.so (MI355X).
 *
 * Same flags, messages and exit codes as the reference CLI (rcanut/nhwcodec encoder/nhw_encoder_cli.c:88-186:
 * -q<N>, -f, -h, -V, <image.bmp> <image.nhw>) and the same BMP acceptance rules as its reader
 * (encoder/nhw_encoder.c:2902-3098: BIH sizes 12/40/52/56/108/124, 512 x +-512, 24 bpp, BI_RGB, bfOffBits
 * honoured, negative height = flipped rows, short pixel data zero-filled).  The pixel work is done by the HIP
 * library through its C ABI (include/nhw_hip.h); this file is host plumbing only.
 *
 * Batch extensions (not in the reference):
 *   nhw-enc [-q N] --batch <dir>                 every <dir>/x.bmp  -> <dir>/x.nhw, one GPU batch per 1024 files
 *   nhw-enc [-q N] --synthetic <count> [--seed S] --outdir <dir>   SURVEY 8d generator on the device
 *   --stock-compat   reproduce the stock one-image-per-process binary instead of the canonical output (include/nhw_hip.h)
 */
#include <dirent.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nhw_hip.h"

#define PROGRAM "nhw-enc"
#define VERSION "0.3.3-mi355x"
#define QUALITY_MIN 0
#define QUALITY_MAX 23
#define QUALITY_DEFAULT 20

/* header check result codes of the reference (nhw_encoder.c:63-71) -- they become process exit codes */
enum { HDR_OK = 0, HDR_NO_DATA = -12, HDR_NO_SIG = -13, HDR_BIH = -14, HDR_PLANES = -15, HDR_FORMAT = -16 };

static void usage(void)
{
	fprintf(stdout,
	        "Usage: %s [-hV][-q<quality>] <image.bmp> <image.nhw>\n"
	        "Convert image: bmp to nwh\n"
	        " (with a bitmap color 512x512 image)\n"
	        "Options:\n"
	        "  -q#       image quality #:[1..23] {default: 20}\n"
	        "  -h        print this help\n"
	        "  -V        show version and legal information\n\n"
	        "  example: nhw-enc -q15 image.bmp image.nhw\n"
	        "Batch (MI355X build): %s [-q#] --batch <dir> | --synthetic <n> [--seed s] --outdir <dir>\n",
	        PROGRAM, PROGRAM);
}

static void version(void)
{
	fprintf(stdout, PROGRAM " " VERSION "\nNHW Image encoder, MI355X-native hot path (from-scratch implementation of the\n"
	                "rcanut/nhwcodec .nhw format; see the repository's license and DESIGN.md).\n");
}

static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

/* returns HDR_*; fills *offset and *flipped */
static int check_header(FILE *f, long *offset, int *flipped)
{
	uint8_t h[34];
	int bih, width, height, planes, bpp, compr;
	if (fseek(f, 0, SEEK_SET) != 0) return -11;
	if (fread(h, 1, sizeof h, f) < sizeof h) return HDR_NO_DATA;
	if (h[0] != 'B' || h[1] != 'M') return HDR_NO_SIG;
	*offset = (long)(int)le32(h + 10);
	bih = (int)le32(h + 14);
	if (bih != 12 && bih != 40 && bih != 52 && bih != 56 && bih != 108 && bih != 124) return HDR_BIH;
	if (bih == 12) { width = le16(h + 18); height = le16(h + 20); planes = (short)le16(h + 22); bpp = (short)le16(h + 24); compr = 0; }
	else { width = (int)le32(h + 18); height = (int)le32(h + 22); planes = (short)le16(h + 26); bpp = (short)le16(h + 28); compr = (int)le32(h + 30); }
	if (planes != 1) return HDR_PLANES;
	if (width != 512 || (height != 512 && height != -512) || bpp != 24 || compr != 0) return HDR_FORMAT;
	*flipped = height < 0;
	return HDR_OK;
}

/* loads one BMP into dst[786432] exactly like the reference's read path; exits like it on errors */
static void load_bmp(const char *path, uint8_t *dst)
{
	FILE *f = fopen(path, "rb");
	long off = 0;
	int flipped = 0, rc, r;
	if (!f) { printf("menu(): Could not open file: %s\n", path); exit(-1); }
	if ((rc = check_header(f, &off, &flipped)) != HDR_OK) { printf("invalid image file.\n"); exit(rc); }
	if (fseek(f, off, SEEK_SET) != 0) { printf("unable to seek to actual data.\n"); exit(-2); }
	memset(dst, 0, NHW_IMG_BYTES);
	if (fread(dst, 1, NHW_IMG_BYTES, f) < NHW_IMG_BYTES) { /* short read tolerated: tail stays zero */ }
	fclose(f);
	if (flipped) {
		uint8_t *tmp = (uint8_t *)malloc(512 * 3);
		for (r = 0; r < 256; r++) {
			memcpy(tmp, dst + (size_t)r * 1536, 1536);
			memcpy(dst + (size_t)r * 1536, dst + (size_t)(511 - r) * 1536, 1536);
			memcpy(dst + (size_t)(511 - r) * 1536, tmp, 1536);
		}
		free(tmp);
	}
}

static int write_file(const char *path, const uint8_t *p, size_t n)
{
	FILE *f = fopen(path, "wb");
	if (!f) { printf("Failed to create file: %s\n", path); return -1; }
	fwrite(p, 1, n, f);
	fclose(f);
	return 0;
}

static void die_lib(const char *what, int rc)
{
	fprintf(stderr, "%s: %s failed (%d): %s\n", PROGRAM, what, rc, nhw_last_error());
	exit(2);
}

static int encode_host_batch(nhw_enc *enc, const uint8_t *imgs, int n, int quality, char **out_names)
{
	uint8_t *arena = (uint8_t *)malloc((size_t)n * NHW_OUT_STRIDE);
	uint64_t *off = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n + 1));
	int32_t *st = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
	int i, rc, bad = 0;
	rc = nhw_enc_batch(enc, imgs, n, quality, arena, (size_t)n * NHW_OUT_STRIDE, off, st);
	if (rc) die_lib("nhw_enc_batch", rc);
	for (i = 0; i < n; i++) {
		if (st[i]) { fprintf(stderr, "%s: %s: encoder status %d (code book overflow)\n", PROGRAM, out_names[i], st[i]); bad++; continue; }
		if (write_file(out_names[i], arena + off[i], (size_t)(off[i + 1] - off[i]))) bad++;
	}
	free(st); free(off); free(arena);
	return bad;
}

static int ends_with(const char *s, const char *suf)
{
	size_t a = strlen(s), b = strlen(suf);
	return a >= b && strcmp(s + a - b, suf) == 0;
}

int main(int argc, char **argv)
{
	int quality = QUALITY_DEFAULT, overwrite = 0, synthetic = 0, i;
	uint32_t seed = 0;
	const char *batch_dir = NULL, *outdir = NULL;
	int stock_compat = 0;   /* --stock-compat: NHW_COMPAT_GLIBC_ONESHOT, the stock binary's out-of-bounds reads (include/nhw_hip.h) */
	nhw_enc *enc = NULL;
	int rc;

	while (argc > 1 && argv[1][0] == '-') {
		if (!strcmp(argv[1], "--batch") && argc > 2) { batch_dir = argv[2]; argc -= 2; argv += 2; continue; }
		if (!strcmp(argv[1], "--synthetic") && argc > 2) { synthetic = atoi(argv[2]); argc -= 2; argv += 2; continue; }
		if (!strcmp(argv[1], "--seed") && argc > 2) { seed = (uint32_t)strtoul(argv[2], NULL, 10); argc -= 2; argv += 2; continue; }
		if (!strcmp(argv[1], "--outdir") && argc > 2) { outdir = argv[2]; argc -= 2; argv += 2; continue; }
		if (!strcmp(argv[1], "--stock-compat")) { stock_compat = 1; argc -= 1; argv += 1; continue; }
		for (i = 1; argv[1][i] != '\0'; i++) {
			const char ch = argv[1][i];
			if (ch >= '0' && ch <= '9') continue;
			if (ch == 'q') {
				const char *num = &argv[1][i + 1];
				if (*num >= '0' && *num <= '9') {
					quality = atoi(num);
					if (quality < QUALITY_MIN || quality > QUALITY_MAX) { printf("quality=%d out of range\n", quality); exit(1); }
				} else { printf("invalid quality='%s'\n", num); exit(1); }
			}
			else if (ch == 'f') overwrite = 1;
			else if (ch == 'h') { usage(); exit(0); }
			else if (ch == 'V') { version(); exit(0); }
			else { fprintf(stderr, "Unknown option '-%c'\n", ch); exit(1); }
		}
		argc--; argv++;
	}
	(void)overwrite; /* the reference's overwrite check is effectively off (its flag is never initialised, nhw_encoder_cli.c:93,164) */

	if (!nhw_quality_supported(quality)) {
		fprintf(stderr, "%s: quality %d is not implemented by the MI355X path in this revision (supported: 17..23)\n", PROGRAM, quality);
		return 3;
	}

	if (synthetic > 0) {
		fprintf(stderr, "%s: --synthetic is served by bench.py / the Python binding (device-resident generator)\n", PROGRAM);
		(void)seed; (void)outdir;
		return 3;
	}

	if (batch_dir) {
		DIR *d = opendir(batch_dir);
		struct dirent *de;
		char **in = NULL, **out = NULL;
		int n = 0, cap = 0, bad = 0, base;
		uint8_t *imgs;
		if (!d) { printf("menu(): Could not open file: %s\n", batch_dir); exit(-1); }
		while ((de = readdir(d))) {
			if (!ends_with(de->d_name, ".bmp")) continue;
			if (n == cap) { cap = cap ? cap * 2 : 256; in = (char **)realloc(in, sizeof(char *) * cap); out = (char **)realloc(out, sizeof(char *) * cap); }
			in[n] = (char *)malloc(strlen(batch_dir) + strlen(de->d_name) + 2);
			sprintf(in[n], "%s/%s", batch_dir, de->d_name);
			out[n] = strdup(in[n]);
			strcpy(out[n] + strlen(out[n]) - 4, ".nhw");
			n++;
		}
		closedir(d);
		if (!n) { printf("Not enough arguments. Check help.\n"); return 0; }
		if ((rc = nhw_enc_create(0, n < 1024 ? n : 1024, &enc))) die_lib("nhw_enc_create", rc);
		if (stock_compat) nhw_enc_set_compat(enc, NHW_COMPAT_GLIBC_ONESHOT);
		imgs = (uint8_t *)malloc((size_t)(n < 1024 ? n : 1024) * NHW_IMG_BYTES);
		for (base = 0; base < n; base += 1024) {
			const int m = n - base < 1024 ? n - base : 1024;
			for (i = 0; i < m; i++) load_bmp(in[base + i], imgs + (size_t)i * NHW_IMG_BYTES);
			bad += encode_host_batch(enc, imgs, m, quality, out + base);
		}
		nhw_enc_destroy(enc);
		return bad ? 1 : 0;
	}

	if (argc < 3) { printf("Not enough arguments. Check help.\n"); usage(); return 0; }
	if (strcmp(argv[1], argv[2]) == 0) { fprintf(stdout, "Input and output are the same file: '%s'.\n", argv[1]); return 1; }
	{
		uint8_t *img = (uint8_t *)malloc(NHW_IMG_BYTES);
		char *names[1];
		int bad;
		load_bmp(argv[1], img);
		if ((rc = nhw_enc_create(0, 1, &enc))) die_lib("nhw_enc_create", rc);
		if (stock_compat) nhw_enc_set_compat(enc, NHW_COMPAT_GLIBC_ONESHOT);
		names[0] = argv[2];
		bad = encode_host_batch(enc, img, 1, quality, names);
		nhw_enc_destroy(enc);
		free(img);
		return bad ? 1 : 0;
	}
}
