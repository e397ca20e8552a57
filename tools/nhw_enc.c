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
 *   nhw-enc [-q N] [--gpus G] --batch <dir>      every <dir>/x.bmp  -> <dir>/x.nhw, GPU batches of up to 1024 files
 *   nhw-enc [-q N] [--gpus G | --devices a,b,..] --synthetic <count> [--seed S] --outdir <dir>
 *                                                SURVEY 8d generator on the device -> <dir>/synth_<seed>.nhw
 *   --stock-compat   reproduce the stock one-image-per-process binary instead of the canonical output (include/nhw_hip.h)
 * --gpus G: images are independent (encoder/nhw_encoder_cli.c:175-183 is a per-image sequence), so the job is a queue of chunks of up
 * to 1024 images; one host thread per GPU, each with its own encoder handle, takes the next chunk until the queue is empty.  The work
 * descriptor {first image, count, quality, seed} lives in this process; between processes (one per GPU under torchrun) it travels by
 * RCCL broadcast: nhwcodec_amd/dist.py, bench.py.
 */
#include <dirent.h>
#include <pthread.h>
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
	        "Batch (MI355X build): %s [-q#] [--gpus g | --devices a,b,..] --batch <dir> | --synthetic <n> [--seed s] --outdir <dir>\n"
	        "Tiles (MI355X build): %s [-q#] --tiles <big.bmp> <stem>   (width, height multiples of 512: one <stem>_y<r>_x<c>.nhw per 512x512 tile)\n"
	        "Tar   (MI355X build): %s [-q#] --tar <in.tar> <out.tar>    (every x.bmp member of a ustar archive -> member x.nhw, in order)\n",
	        PROGRAM, PROGRAM, PROGRAM, PROGRAM);
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

/* --tiles: a 24-bit BMP whose sides are multiples of 512 is cut into independent 512x512 tiles, every tile encoded exactly as
 * the reference would encode that crop saved as a BMP of its own (file row order kept; a top-down file is flipped as a whole first, as
 * the reference does with a single tile).  Returns the tiles (malloc), tile (r, c) at index r * nx + c. */
static uint8_t *load_bmp_tiles(const char *path, int *ny, int *nx)
{
	FILE *f = fopen(path, "rb");
	uint8_t h[34], *row, *tiles;
	int bih, width, height, planes, bpp, compr, flipped, r, c;
	long off;
	if (!f) { printf("menu(): Could not open file: %s\n", path); exit(-1); }
	if (fread(h, 1, sizeof h, f) < sizeof h) { printf("invalid image file.\n"); exit(HDR_NO_DATA); }
	if (h[0] != 'B' || h[1] != 'M') { printf("invalid image file.\n"); exit(HDR_NO_SIG); }
	off = (long)(int)le32(h + 10);
	bih = (int)le32(h + 14);
	if (bih != 40 && bih != 52 && bih != 56 && bih != 108 && bih != 124) { printf("invalid image file.\n"); exit(HDR_BIH); }
	width = (int)le32(h + 18); height = (int)le32(h + 22); planes = (short)le16(h + 26); bpp = (short)le16(h + 28); compr = (int)le32(h + 30);
	if (planes != 1) { printf("invalid image file.\n"); exit(HDR_PLANES); }
	flipped = height < 0;
	if (flipped) height = -height;
	if (width < 512 || height < 512 || width % 512 || height % 512 || bpp != 24 || compr != 0) {
		printf("invalid image file.\n");
		fprintf(stderr, "%s: --tiles wants a 24-bit uncompressed BMP whose width and height are multiples of 512 (got %dx%d, %d bpp)\n", PROGRAM, width, height, bpp);
		exit(HDR_FORMAT);
	}
	if (fseek(f, off, SEEK_SET) != 0) { printf("unable to seek to actual data.\n"); exit(-2); }
	*nx = width / 512; *ny = height / 512;
	tiles = (uint8_t *)calloc((size_t)*nx * *ny, NHW_IMG_BYTES);
	row = (uint8_t *)malloc((size_t)width * 3);
	for (r = 0; r < height; r++) {
		const int fr = flipped ? height - 1 - r : r;                  /* row of the picture as the encoder sees it */
		if (fread(row, 1, (size_t)width * 3, f) < (size_t)width * 3) break;   /* short file: the rest stays zero, like the reference's read */
		for (c = 0; c < *nx; c++)
			memcpy(tiles + ((size_t)(fr / 512) * *nx + c) * NHW_IMG_BYTES + (size_t)(fr % 512) * 1536, row + (size_t)c * 1536, 1536);
	}
	free(row);
	fclose(f);
	return tiles;
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

/* the job queue of the batch modes: chunks of up to 1024 images, taken in order by one host thread per GPU */
#define CHUNK 1024                 /* images per GPU batch at most */
static int g_chunk = CHUNK;        /* --chunk N: images a worker takes from the queue at a time (1 .. CHUNK) */
struct job {
	int n, next;                   /* images in the job, first image nobody has taken yet */
	int quality, stock_compat;
	uint32_t seed;                 /* --synthetic: image i is generator seed `seed + i` */
	char **in, **out;              /* --batch: file names */
	const char *outdir;            /* --synthetic */
	pthread_mutex_t lock;
};
struct worker { struct job *jb; int device, bad; };

static void *worker_main(void *arg)
{
	struct worker *w = (struct worker *)arg;
	struct job *jb = w->jb;
	nhw_enc *enc = NULL;
	uint8_t *imgs = NULL, *arena = NULL;
	int imgs_pinned = 0;
	uint64_t *off = (uint64_t *)malloc(sizeof(uint64_t) * (CHUNK + 1));
	int32_t *st = (int32_t *)malloc(sizeof(int32_t) * CHUNK);
	const int cap = jb->n < g_chunk ? jb->n : g_chunk;
	int rc, i;
	if ((rc = nhw_enc_create(w->device, cap, &enc))) die_lib("nhw_enc_create", rc);
	if (jb->stock_compat) nhw_enc_set_compat(enc, NHW_COMPAT_GLIBC_ONESHOT);
	arena = (uint8_t *)malloc((size_t)cap * NHW_OUT_STRIDE);
	if (!jb->outdir) {                                 /* page-locked input buffer: the upload runs at PCIe speed next to the encode */
		imgs = (uint8_t *)nhw_host_alloc((size_t)cap * NHW_IMG_BYTES);
		imgs_pinned = imgs != NULL;
		if (!imgs) imgs = (uint8_t *)malloc((size_t)cap * NHW_IMG_BYTES);
		if (!imgs) { fprintf(stderr, "%s: out of memory for %d input images\n", PROGRAM, cap); exit(-1); }
	}
	for (;;) {
		int base, m;
		pthread_mutex_lock(&jb->lock);
		base = jb->next; m = jb->n - base < g_chunk ? jb->n - base : g_chunk; jb->next += m > 0 ? m : 0;
		pthread_mutex_unlock(&jb->lock);
		if (m <= 0) break;
		if (jb->outdir) rc = nhw_enc_synth_batch(enc, m, jb->seed + (uint32_t)base, jb->quality, arena, (size_t)cap * NHW_OUT_STRIDE, off, st);
		else {
			for (i = 0; i < m; i++) load_bmp(jb->in[base + i], imgs + (size_t)i * NHW_IMG_BYTES);
			rc = nhw_enc_batch(enc, imgs, m, jb->quality, arena, (size_t)cap * NHW_OUT_STRIDE, off, st);
		}
		if (rc) die_lib("encode", rc);
		for (i = 0; i < m; i++) {
			char name[4096];
			const char *path = name;
			if (jb->outdir) snprintf(name, sizeof name, "%s/synth_%u.nhw", jb->outdir, (unsigned)(jb->seed + (uint32_t)(base + i)));
			else path = jb->out[base + i];
			if (st[i]) { fprintf(stderr, "%s: %s: encoder status %d (code book overflow)\n", PROGRAM, path, st[i]); w->bad++; continue; }
			if (write_file(path, arena + off[i], (size_t)(off[i + 1] - off[i]))) w->bad++;
		}
	}
	if (imgs_pinned) nhw_host_free(imgs); else free(imgs);        /* before the handle goes: page-locked memory belongs to its device context */
	nhw_enc_destroy(enc);
	free(st); free(off); free(arena);
	return NULL;
}

/* --tar: a POSIX ustar archive as the batch container.  Members are read in order; every regular member whose name ends in
 * .bmp goes through the same header checks as a file on its own (a member that fails them is reported and left out), the images are
 * encoded in chunks of CHUNK, and the .nhw files leave as members of the output archive under the same names with the suffix changed. */
static unsigned long tar_octal(const uint8_t *p, int n) { unsigned long v = 0; int i; for (i = 0; i < n && p[i] >= '0' && p[i] <= '7'; i++) v = v * 8 + (unsigned long)(p[i] - '0'); return v; }
static int tar_write_member(FILE *f, const char *name, const uint8_t *data, size_t len)
{
	uint8_t h[512];
	unsigned sum = 0;
	size_t i, pad = (512 - len % 512) % 512;
	static const uint8_t zeros[512];
	memset(h, 0, sizeof h);
	if (strlen(name) > 99) { fprintf(stderr, "%s: member name too long for the archive: %s\n", PROGRAM, name); return -1; }
	strcpy((char *)h, name);
	memcpy(h + 100, "0000644", 8); memcpy(h + 108, "0000000", 8); memcpy(h + 116, "0000000", 8);
	snprintf((char *)h + 124, 12, "%011lo", (unsigned long)len);
	memcpy(h + 136, "00000000000", 12);
	memset(h + 148, ' ', 8);
	h[156] = '0';
	memcpy(h + 257, "ustar", 6); memcpy(h + 263, "00", 2);
	for (i = 0; i < 512; i++) sum += h[i];
	snprintf((char *)h + 148, 8, "%06o", sum); h[155] = ' ';
	if (fwrite(h, 1, 512, f) != 512 || fwrite(data, 1, len, f) != len || fwrite(zeros, 1, pad, f) != pad) return -1;
	return 0;
}
/* a BMP held in memory -> dst[786432], the same acceptance rules as check_header + load_bmp; returns HDR_OK or the reference's code */
static int bmp_from_memory(const uint8_t *b, size_t len, uint8_t *dst)
{
	int bih, width, height, planes, bpp, compr, r;
	long off;
	if (len < 34) return HDR_NO_DATA;
	if (b[0] != 'B' || b[1] != 'M') return HDR_NO_SIG;
	off = (long)(int)le32(b + 10);
	bih = (int)le32(b + 14);
	if (bih != 12 && bih != 40 && bih != 52 && bih != 56 && bih != 108 && bih != 124) return HDR_BIH;
	if (bih == 12) { width = le16(b + 18); height = le16(b + 20); planes = (short)le16(b + 22); bpp = (short)le16(b + 24); compr = 0; }
	else { width = (int)le32(b + 18); height = (int)le32(b + 22); planes = (short)le16(b + 26); bpp = (short)le16(b + 28); compr = (int)le32(b + 30); }
	if (planes != 1) return HDR_PLANES;
	if (width != 512 || (height != 512 && height != -512) || bpp != 24 || compr != 0) return HDR_FORMAT;
	memset(dst, 0, NHW_IMG_BYTES);
	if (off >= 0 && (size_t)off < len) memcpy(dst, b + off, len - (size_t)off < NHW_IMG_BYTES ? len - (size_t)off : NHW_IMG_BYTES);   /* short data: the tail stays zero */
	if (height < 0)
		for (r = 0; r < 256; r++) {
			uint8_t tmp[1536];
			memcpy(tmp, dst + (size_t)r * 1536, 1536);
			memcpy(dst + (size_t)r * 1536, dst + (size_t)(511 - r) * 1536, 1536);
			memcpy(dst + (size_t)(511 - r) * 1536, tmp, 1536);
		}
	return HDR_OK;
}
#define TAR_BMP_MAX (4u << 20)      /* largest archive member taken for an image */
/* skip n bytes of the archive: by seeking, or by reading where the input cannot seek (a pipe); non-zero at the end of the input */
static int tar_skip(FILE *in, unsigned long n)
{
	uint8_t sink[4096];
	if (n == 0) return 0;
	if (fseek(in, (long)n, SEEK_CUR) == 0) return 0;
	while (n) { const size_t k = n < sizeof sink ? (size_t)n : sizeof sink; if (fread(sink, 1, k, in) != k) return 1; n -= (unsigned long)k; }
	return 0;
}
static int encode_tar(const char *in_path, const char *out_path, int quality, int stock_compat)
{
	FILE *in = fopen(in_path, "rb"), *out;
	nhw_enc *enc = NULL;
	uint8_t hdr[512], *imgs, *arena, *member = NULL;
	size_t member_cap = 0;
	uint64_t *off = (uint64_t *)malloc(sizeof(uint64_t) * (CHUNK + 1));
	int32_t *st = (int32_t *)malloc(sizeof(int32_t) * CHUNK);
	char (*names)[104] = (char (*)[104])malloc((size_t)CHUNK * 104);
	int n = 0, total = 0, bad = 0, rc, i, eof = 0;
	static const uint8_t zeros[1024];
	if (!in) { printf("menu(): Could not open file: %s\n", in_path); exit(-1); }
	out = fopen(out_path, "wb");
	if (!out) { printf("Failed to create file: %s\n", out_path); return 1; }
	imgs = (uint8_t *)malloc((size_t)CHUNK * NHW_IMG_BYTES);
	arena = (uint8_t *)malloc((size_t)CHUNK * NHW_OUT_STRIDE);
	while (!eof) {
		unsigned long size = 0;
		int is_bmp = 0;
		if (fread(hdr, 1, 512, in) != 512 || hdr[0] == 0) eof = 1;          /* end of archive: a zero block (or the file's end) */
		else {
			size_t nl;
			size = tar_octal(hdr + 124, 12);
			hdr[99] = 0;
			nl = strlen((const char *)hdr);
			is_bmp = (hdr[156] == '0' || hdr[156] == 0) && nl > 4 && !strcmp((const char *)hdr + nl - 4, ".bmp");
			if (is_bmp) {
				if (size > TAR_BMP_MAX) {                          /* a 512 x 512 24-bit BMP is 786 486 bytes; a header that claims gigabytes is damage */
					fprintf(stderr, "%s: %s: member of %lu bytes is no 512 x 512 image, left out\n", PROGRAM, (const char *)hdr, size); bad++;
					if (tar_skip(in, (size + 511) / 512 * 512)) { fprintf(stderr, "%s: %s: archive ends inside member %s\n", PROGRAM, in_path, (const char *)hdr); eof = 1; }
					is_bmp = 0;
				}
				else if (size + 1 > member_cap) {
					uint8_t *grown = (uint8_t *)realloc(member, size + 1);
					if (!grown) { fprintf(stderr, "%s: out of memory for a member of %lu bytes\n", PROGRAM, size); exit(-1); }
					member = grown; member_cap = size + 1;
				}
				if (!is_bmp) { }
				else if (fread(member, 1, size, in) != size) { fprintf(stderr, "%s: %s: archive ends inside member %s\n", PROGRAM, in_path, (const char *)hdr); bad++; eof = 1; is_bmp = 0; }
				else {
					const int hc = bmp_from_memory(member, size, imgs + (size_t)n * NHW_IMG_BYTES);
					if (hc != HDR_OK) { fprintf(stderr, "%s: %s: invalid image file (%d), left out\n", PROGRAM, (const char *)hdr, hc); bad++; }
					else { snprintf(names[n], 104, "%.*s.nhw", (int)(nl - 4), (const char *)hdr); n++; }
				}
				if (is_bmp && size % 512 && tar_skip(in, 512 - size % 512)) eof = 1;
			}
			else if (tar_skip(in, (size + 511) / 512 * 512)) eof = 1;
		}
		if (n == CHUNK || (eof && n > 0)) {
			if (!enc) {
				if ((rc = nhw_enc_create(0, CHUNK, &enc))) die_lib("nhw_enc_create", rc);
				if (stock_compat) nhw_enc_set_compat(enc, NHW_COMPAT_GLIBC_ONESHOT);
			}
			if ((rc = nhw_enc_batch(enc, imgs, n, quality, arena, (size_t)CHUNK * NHW_OUT_STRIDE, off, st))) die_lib("nhw_enc_batch", rc);
			for (i = 0; i < n; i++) {
				if (st[i]) { fprintf(stderr, "%s: %s: encoder status %d (code book overflow)\n", PROGRAM, names[i], st[i]); bad++; continue; }
				if (tar_write_member(out, names[i], arena + off[i], (size_t)(off[i + 1] - off[i]))) { bad++; break; }
			}
			total += n; n = 0;
		}
	}
	fwrite(zeros, 1, 1024, out);
	fclose(out); fclose(in);
	if (enc) nhw_enc_destroy(enc);
	printf("%d image(s) encoded\n", total);
	free(member); free(names); free(st); free(off); free(arena); free(imgs);
	return bad ? 1 : 0;
}

int main(int argc, char **argv)
{
	int quality = QUALITY_DEFAULT, overwrite = 0, synthetic = 0, gpus = 1, i;
	int devlist[16], ndevlist = 0;
	uint32_t seed = 0;
	const char *batch_dir = NULL, *outdir = NULL;
	int tiles = 0, tar = 0;
	int stock_compat = 0;   /* --stock-compat: NHW_COMPAT_GLIBC_ONESHOT, the stock binary's out-of-bounds reads (include/nhw_hip.h) */
	nhw_enc *enc = NULL;
	int rc;

	while (argc > 1 && argv[1][0] == '-') {
		if (!strcmp(argv[1], "--batch") && argc > 2) { batch_dir = argv[2]; argc -= 2; argv += 2; continue; }
		if (!strcmp(argv[1], "--synthetic") && argc > 2) { synthetic = atoi(argv[2]); argc -= 2; argv += 2; continue; }
		if (!strcmp(argv[1], "--seed") && argc > 2) { seed = (uint32_t)strtoul(argv[2], NULL, 10); argc -= 2; argv += 2; continue; }
		if (!strcmp(argv[1], "--outdir") && argc > 2) { outdir = argv[2]; argc -= 2; argv += 2; continue; }
		if (!strcmp(argv[1], "--gpus") && argc > 2) { gpus = atoi(argv[2]); argc -= 2; argv += 2; continue; }
		if (!strcmp(argv[1], "--chunk") && argc > 2) {
			char *end;
			const long v = strtol(argv[2], &end, 10);
			if (end == argv[2] || *end || v < 1 || v > CHUNK) { fprintf(stderr, "%s: --chunk wants a number 1..%d, got '%s'\n", PROGRAM, CHUNK, argv[2]); return 1; }
			g_chunk = (int)v; argc -= 2; argv += 2; continue;
		}
		if (!strcmp(argv[1], "--devices") && argc > 2) {            /* one worker per entry; a device may be named more than once (two handles, two workers on one GPU) */
			const char *p = argv[2];
			for (;;) {                                                /* digits, comma, digits, ..: anything else (empty list, trailing comma, garbage, more than 16 entries) is an error, not device 0 */
				char *end;
				const long v = strtol(p, &end, 10);
				if (end == p || v < 0 || v > 1023) { fprintf(stderr, "%s: --devices wants a comma-separated list of device numbers, got '%s'\n", PROGRAM, argv[2]); return 1; }
				if (ndevlist >= 16) { fprintf(stderr, "%s: --devices: at most 16 entries\n", PROGRAM); return 1; }
				devlist[ndevlist++] = (int)v;
				if (*end == '\0') break;
				if (*end != ',' || end[1] == '\0') { fprintf(stderr, "%s: --devices wants a comma-separated list of device numbers, got '%s'\n", PROGRAM, argv[2]); return 1; }
				p = end + 1;
			}
			argc -= 2; argv += 2; continue;
		}
		if (!strcmp(argv[1], "--stock-compat")) { stock_compat = 1; argc -= 1; argv += 1; continue; }
		if (!strcmp(argv[1], "--tiles")) { tiles = 1; argc -= 1; argv += 1; continue; }
		if (!strcmp(argv[1], "--tar")) { tar = 1; argc -= 1; argv += 1; continue; }
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
		fprintf(stderr, "%s: quality %d is not implemented (supported: 1..23; the reference accepts -q0 but has no tables for it)\n", PROGRAM, quality);
		return 3;
	}

	if (synthetic > 0 || batch_dir) {
		struct job jb;
		pthread_t th[16];
		struct worker wk[16];
		int ndev = nhw_device_count(), g, bad = 0;
		memset(&jb, 0, sizeof jb);
		jb.quality = quality; jb.stock_compat = stock_compat; jb.seed = seed;
		pthread_mutex_init(&jb.lock, NULL);
		if (synthetic > 0) {
			if (!outdir) { fprintf(stderr, "%s: --synthetic needs --outdir <dir>\n", PROGRAM); return 1; }
			jb.n = synthetic; jb.outdir = outdir;
		} else {
			DIR *d = opendir(batch_dir);
			struct dirent *de;
			int cap = 0;
			if (!d) { printf("menu(): Could not open file: %s\n", batch_dir); exit(-1); }
			while ((de = readdir(d))) {
				if (!ends_with(de->d_name, ".bmp")) continue;
				if (jb.n == cap) { cap = cap ? cap * 2 : 256; jb.in = (char **)realloc(jb.in, sizeof(char *) * cap); jb.out = (char **)realloc(jb.out, sizeof(char *) * cap); }
				jb.in[jb.n] = (char *)malloc(strlen(batch_dir) + strlen(de->d_name) + 2);
				sprintf(jb.in[jb.n], "%s/%s", batch_dir, de->d_name);
				jb.out[jb.n] = strdup(jb.in[jb.n]);
				strcpy(jb.out[jb.n] + strlen(jb.out[jb.n]) - 4, ".nhw");
				jb.n++;
			}
			closedir(d);
			if (!jb.n) { printf("Not enough arguments. Check help.\n"); return 0; }
		}
		if (ndev < 1) { fprintf(stderr, "%s: no GPU visible\n", PROGRAM); return 2; }
		if (gpus < 1) gpus = 1;
		if (ndevlist) {
			gpus = ndevlist;
			for (g = 0; g < gpus; g++) if (devlist[g] < 0 || devlist[g] >= ndev) { fprintf(stderr, "%s: --devices names device %d but %d device(s) visible\n", PROGRAM, devlist[g], ndev); return 1; }
		} else {
			if (gpus > ndev) { fprintf(stderr, "%s: --gpus %d but %d device(s) visible\n", PROGRAM, gpus, ndev); return 1; }
			if (gpus > 16) gpus = 16;
			for (g = 0; g < gpus; g++) devlist[g] = g;
		}
		for (g = 0; g < gpus; g++) { wk[g].jb = &jb; wk[g].device = devlist[g]; wk[g].bad = 0; pthread_create(&th[g], NULL, worker_main, &wk[g]); }
		for (g = 0; g < gpus; g++) { pthread_join(th[g], NULL); bad += wk[g].bad; }
		return bad ? 1 : 0;
	}

	if (argc < 3) { printf("Not enough arguments. Check help.\n"); usage(); return 0; }
	if (tar) return encode_tar(argv[1], argv[2], quality, stock_compat);
	if (tiles) {
		int ny = 0, nx = 0, n, bad, t;
		uint8_t *imgs = load_bmp_tiles(argv[1], &ny, &nx);
		char **names;
		n = ny * nx;
		names = (char **)malloc(sizeof(char *) * (size_t)n);
		for (t = 0; t < n; t++) {
			names[t] = (char *)malloc(strlen(argv[2]) + 48);
			sprintf(names[t], "%s_y%d_x%d.nhw", argv[2], t / nx, t % nx);
		}
		if ((rc = nhw_enc_create(0, n, &enc))) die_lib("nhw_enc_create", rc);
		if (stock_compat) nhw_enc_set_compat(enc, NHW_COMPAT_GLIBC_ONESHOT);
		bad = encode_host_batch(enc, imgs, n, quality, names);
		nhw_enc_destroy(enc);
		printf("%d x %d tiles\n", ny, nx);
		free(imgs);
		return bad ? 1 : 0;
	}
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
