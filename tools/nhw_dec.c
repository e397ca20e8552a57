/*
 * nhw-dec -- command-line decoder over libnhwhip.so (include/nhw_hip.h).
 *
 * Same interface as the reference CLI (rcanut/nhwcodec decoder/nhw_decoder_cli.c:70-118):
 *     nhw-dec image.nhw image.bmp          (with fewer arguments: the usage text, exit 0)
 * The BMP is the 54-byte header of decoder/nhw_decoder_cli.c:61-65,293-312 followed by the pixel bytes in the order
 * write_image_bmp (:108-291) writes them.  Exit codes: 0 ok, 1 cannot read / write a file (the reference prints and
 * carries on into undefined behaviour), 3 not an .nhw file (reference: "Not an .nhw file", exit(-1)).
 * Extension: --batch <dir> decodes every *.nhw of a directory to <name>.bmp in one GPU batch.
 */
#include <dirent.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/nhw_hip.h"

#define PROGRAM "nhw-dec"

static void show_usage(void)
{
	fprintf(stdout,
	"Usage: %s <image.nhw> <image.bmp>\n"
	"Convert image: nwh to bmp\n"
	" (with a bitmap color 512x512 image)\n"
	"\n"
	"  example: nhw-dec image.nhw image.bmp\n"
	"  batch:   nhw-dec --batch <directory of .nhw files>\n"
	"  tiles:   nhw-dec --tiles <rows> <columns> <stem> <image.bmp>   (joins <stem>_y<r>_x<c>.nhw, as written by nhw-enc --tiles)\n"
	"  tar:     nhw-dec --tar <in.tar> <out.tar>   (every x.nhw member of a ustar archive -> member x.bmp, in order)\n",
	PROGRAM);
}

static int read_file(const char *path, uint8_t **buf, size_t *len)
{
	FILE *f = fopen(path, "rb");
	long n;
	if (!f) { printf("\nCould not open file\n"); return 1; }
	fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
	*buf = (uint8_t *)malloc((size_t)n + 1);
	if (!*buf || fread(*buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); return 1; }
	fclose(f);
	*len = (size_t)n;
	return 0;
}

static int write_bmp(const char *path, const uint8_t *pixels)
{
	uint8_t hdr[54];
	FILE *f = fopen(path, "wb");
	if (!f) { printf("Failed to open output decompressed .bmp file %s\n", path); return 1; }
	nhw_dec_bmp_header(hdr);
	fwrite(hdr, 54, 1, f);
	fwrite(pixels, NHW_IMG_BYTES, 1, f);
	fclose(f);
	return 0;
}

static int decode_files(char **in, char **out, int n)
{
	nhw_dec *d = NULL;
	uint8_t *blob = NULL, *pix;
	uint64_t *off = (uint64_t *)calloc((size_t)n + 1, sizeof *off);
	int32_t *status = (int32_t *)calloc((size_t)n, sizeof *status);
	size_t total = 0;
	int i, rc = 0;
	for (i = 0; i < n; i++) {
		uint8_t *b; size_t len;
		if (read_file(in[i], &b, &len)) return 1;
		blob = (uint8_t *)realloc(blob, total + len + 16);
		memcpy(blob + total, b, len); free(b);
		off[i] = total; total += len;
	}
	off[n] = total;
	pix = (uint8_t *)malloc((size_t)n * NHW_IMG_BYTES);
	if (nhw_dec_create(0, n, &d) || nhw_dec_batch(d, blob, off, n, pix, status, NULL)) {
		fprintf(stderr, "%s: GPU decoder unavailable: %s\n", PROGRAM, nhw_dec_last_error());
		return 2;
	}
	for (i = 0; i < n; i++) {
		if (status[i]) { printf("\nNot an .nhw file"); if (n > 1) printf(": %s", in[i]); printf("\n"); rc = 3; continue; }
		if (write_bmp(out[i], pix + (size_t)i * NHW_IMG_BYTES)) rc = rc ? rc : 1;
	}
	nhw_dec_destroy(d);
	free(blob); free(pix); free(off); free(status);
	return rc;
}

/* --tiles: the inverse of nhw-enc --tiles.  Every tile is a file of its own and decodes on its own; the tiles are put side by side in file
 * row order under one bottom-up BMP header of the whole size. */
static int decode_tiles(int ny, int nx, const char *stem, const char *out_path)
{
	const int n = ny * nx;
	nhw_dec *d = NULL;
	uint8_t *blob = NULL, *pix, hdr[54];
	uint64_t *off = (uint64_t *)calloc((size_t)n + 1, sizeof *off);
	int32_t *status = (int32_t *)calloc((size_t)n, sizeof *status);
	size_t total = 0;
	const uint32_t width = 512u * (uint32_t)nx, height = 512u * (uint32_t)ny, bytes = width * height * 3u;
	FILE *f;
	int t, r;
	if ((uint64_t)width * height * 3u + 54u > 0xFFFFFFFFull) {       /* the BMP header's 32-bit size fields */
		fprintf(stderr, "%s: a %d x %d grid of tiles does not fit a BMP file (4 GiB)\n", PROGRAM, ny, nx);
		return 1;
	}
	for (t = 0; t < n; t++) {
		char name[4096];
		uint8_t *b; size_t len;
		snprintf(name, sizeof name, "%s_y%d_x%d.nhw", stem, t / nx, t % nx);
		if (read_file(name, &b, &len)) return 1;
		{ uint8_t *grown = (uint8_t *)realloc(blob, total + len + 16);
		  if (!grown) { fprintf(stderr, "%s: out of memory\n", PROGRAM); return 1; }
		  blob = grown; }
		memcpy(blob + total, b, len); free(b);
		off[t] = total; total += len;
	}
	off[n] = total;
	pix = (uint8_t *)malloc((size_t)n * NHW_IMG_BYTES);
	if (nhw_dec_create(0, n, &d) || nhw_dec_batch(d, blob, off, n, pix, status, NULL)) {
		fprintf(stderr, "%s: GPU decoder unavailable: %s\n", PROGRAM, nhw_dec_last_error());
		return 2;
	}
	for (t = 0; t < n; t++) if (status[t]) { printf("\nNot an .nhw file: %s_y%d_x%d.nhw\n", stem, t / nx, t % nx); return 3; }
	nhw_dec_bmp_header(hdr);                                          /* the reference's 54 bytes, with the size fields of the whole picture */
	hdr[2] = (uint8_t)(bytes + 54); hdr[3] = (uint8_t)((bytes + 54) >> 8); hdr[4] = (uint8_t)((bytes + 54) >> 16); hdr[5] = (uint8_t)((bytes + 54) >> 24);
	hdr[18] = (uint8_t)width; hdr[19] = (uint8_t)(width >> 8); hdr[20] = (uint8_t)(width >> 16); hdr[21] = (uint8_t)(width >> 24);
	hdr[22] = (uint8_t)height; hdr[23] = (uint8_t)(height >> 8); hdr[24] = (uint8_t)(height >> 16); hdr[25] = (uint8_t)(height >> 24);
	hdr[34] = (uint8_t)bytes; hdr[35] = (uint8_t)(bytes >> 8); hdr[36] = (uint8_t)(bytes >> 16); hdr[37] = (uint8_t)(bytes >> 24);
	f = fopen(out_path, "wb");
	if (!f) { printf("Failed to open output decompressed .bmp file %s\n", out_path); return 1; }
	fwrite(hdr, 54, 1, f);
	for (r = 0; r < (int)height; r++)
		for (t = 0; t < nx; t++)
			fwrite(pix + ((size_t)(r / 512) * nx + t) * NHW_IMG_BYTES + (size_t)(r % 512) * 1536, 1536, 1, f);
	fclose(f);
	nhw_dec_destroy(d);
	free(blob); free(pix); free(off); free(status);
	printf("%d x %d tiles\n", ny, nx);
	return 0;
}

/* --tar: the inverse of nhw-enc --tar; members are decoded in batches of up to 1024 */
static unsigned long tar_octal(const uint8_t *p, int n) { unsigned long v = 0; int i; for (i = 0; i < n && p[i] >= '0' && p[i] <= '7'; i++) v = v * 8 + (unsigned long)(p[i] - '0'); return v; }
static int tar_write_member(FILE *f, const char *name, const uint8_t *head, size_t head_len, const uint8_t *data, size_t len)
{
	uint8_t h[512];
	unsigned sum = 0;
	const size_t total = head_len + len, pad = (512 - total % 512) % 512;
	size_t i;
	static const uint8_t zeros[512];
	memset(h, 0, sizeof h);
	if (strlen(name) > 99) return -1;
	strcpy((char *)h, name);
	memcpy(h + 100, "0000644", 8); memcpy(h + 108, "0000000", 8); memcpy(h + 116, "0000000", 8);
	snprintf((char *)h + 124, 12, "%011lo", (unsigned long)total);
	memcpy(h + 136, "00000000000", 12);
	memset(h + 148, ' ', 8);
	h[156] = '0';
	memcpy(h + 257, "ustar", 6); memcpy(h + 263, "00", 2);
	for (i = 0; i < 512; i++) sum += h[i];
	snprintf((char *)h + 148, 8, "%06o", sum); h[155] = ' ';
	return (fwrite(h, 1, 512, f) == 512 && fwrite(head, 1, head_len, f) == head_len && fwrite(data, 1, len, f) == len && fwrite(zeros, 1, pad, f) == pad) ? 0 : -1;
}
#define TAR_NHW_MAX (1u << 20)      /* largest archive member taken for a .nhw file */
/* skip n bytes of the archive: by seeking, or by reading where the input cannot seek (a pipe); non-zero at the end of the input */
static int tar_skip(FILE *in, unsigned long n)
{
	uint8_t sink[4096];
	if (n == 0) return 0;
	if (fseek(in, (long)n, SEEK_CUR) == 0) return 0;
	while (n) { const size_t k = n < sizeof sink ? (size_t)n : sizeof sink; if (fread(sink, 1, k, in) != k) return 1; n -= (unsigned long)k; }
	return 0;
}
static int decode_tar(const char *in_path, const char *out_path)
{
	enum { CH = 1024 };
	FILE *in = fopen(in_path, "rb"), *out;
	nhw_dec *d = NULL;
	uint8_t hdr[512], bmp[54], *blob = NULL, *pix;
	size_t blob_cap = 0, total_bytes = 0;
	uint64_t *off = (uint64_t *)calloc(CH + 1, sizeof *off);
	int32_t *status = (int32_t *)calloc(CH, sizeof *status);
	char (*names)[104] = (char (*)[104])malloc((size_t)CH * 104);
	int n = 0, total = 0, bad = 0, eof = 0, i;
	static const uint8_t zeros[1024];
	if (!in) { printf("\nCould not open file\n"); return 1; }
	out = fopen(out_path, "wb");
	if (!out) { printf("Failed to open output decompressed .bmp file %s\n", out_path); return 1; }
	pix = (uint8_t *)malloc((size_t)CH * NHW_IMG_BYTES);
	nhw_dec_bmp_header(bmp);
	while (!eof) {
		if (fread(hdr, 1, 512, in) != 512 || hdr[0] == 0) eof = 1;
		else {
			const unsigned long size = tar_octal(hdr + 124, 12);
			size_t nl;
			hdr[99] = 0;
			nl = strlen((const char *)hdr);
			if ((hdr[156] == '0' || hdr[156] == 0) && nl > 4 && !strcmp((const char *)hdr + nl - 4, ".nhw")) {
				if (size > TAR_NHW_MAX) {                          /* the encoder never writes more than 512 KiB a file */
					fprintf(stderr, "%s: %s: member of %lu bytes is no .nhw file, left out\n", PROGRAM, (const char *)hdr, size); bad++;
					if (tar_skip(in, (size + 511) / 512 * 512)) eof = 1;
				}
				else {
				if (total_bytes + size + 16 > blob_cap) {
					uint8_t *grown = (uint8_t *)realloc(blob, (total_bytes + size + 16) * 2);
					if (!grown) { fprintf(stderr, "%s: out of memory\n", PROGRAM); return 1; }
					blob = grown; blob_cap = (total_bytes + size + 16) * 2;
				}
				if (fread(blob + total_bytes, 1, size, in) != size) { fprintf(stderr, "%s: %s: archive ends inside member %s\n", PROGRAM, in_path, (const char *)hdr); bad++; eof = 1; }
				else { off[n] = total_bytes; total_bytes += size; snprintf(names[n], 104, "%.*s.bmp", (int)(nl - 4), (const char *)hdr); n++; }
				if (size % 512 && tar_skip(in, 512 - size % 512)) eof = 1;
				}
			}
			else if (tar_skip(in, (size + 511) / 512 * 512)) eof = 1;
		}
		if (n == CH || (eof && n > 0)) {
			off[n] = total_bytes;
			if ((!d && nhw_dec_create(0, CH, &d)) || nhw_dec_batch(d, blob, off, n, pix, status, NULL)) {
				fprintf(stderr, "%s: GPU decoder unavailable: %s\n", PROGRAM, nhw_dec_last_error());
				return 2;
			}
			for (i = 0; i < n; i++) {
				if (status[i]) { printf("\nNot an .nhw file: %s\n", names[i]); bad++; continue; }
				if (tar_write_member(out, names[i], bmp, 54, pix + (size_t)i * NHW_IMG_BYTES, NHW_IMG_BYTES)) { bad++; break; }
			}
			total += n; n = 0; total_bytes = 0;
		}
	}
	fwrite(zeros, 1, 1024, out);
	fclose(out); fclose(in);
	if (d) nhw_dec_destroy(d);
	printf("%d file(s) decoded\n", total);
	free(blob); free(pix); free(off); free(status); free(names);
	return bad ? 3 : 0;
}

int main(int argc, char **argv)
{
	if (argc < 3) { show_usage(); return 0; }
	if (!strcmp(argv[1], "--tar")) {
		if (argc < 4) { show_usage(); return 1; }
		return decode_tar(argv[2], argv[3]);
	}
	if (!strcmp(argv[1], "--tiles")) {
		int ny, nx;
		if (argc < 6 || (ny = atoi(argv[2])) < 1 || (nx = atoi(argv[3])) < 1 || ny * nx > 65535) { show_usage(); return 1; }
		return decode_tiles(ny, nx, argv[4], argv[5]);
	}
	if (!strcmp(argv[1], "--batch")) {
		DIR *dir = opendir(argv[2]);
		struct dirent *e;
		char **in = NULL, **out = NULL;
		int n = 0, rc;
		if (!dir) { printf("\nCould not open file\n"); return 1; }
		while ((e = readdir(dir))) {
			const size_t l = strlen(e->d_name);
			if (l < 5 || strcmp(e->d_name + l - 4, ".nhw")) continue;
			in = (char **)realloc(in, (size_t)(n + 1) * sizeof *in); out = (char **)realloc(out, (size_t)(n + 1) * sizeof *out);
			in[n] = (char *)malloc(strlen(argv[2]) + l + 2); out[n] = (char *)malloc(strlen(argv[2]) + l + 2);
			sprintf(in[n], "%s/%s", argv[2], e->d_name);
			sprintf(out[n], "%s/%.*s.bmp", argv[2], (int)(l - 4), e->d_name);
			n++;
		}
		closedir(dir);
		if (!n) return 0;
		rc = decode_files(in, out, n);
		printf("%d file(s) decoded\n", n);
		return rc;
	}
	return decode_files(&argv[1], &argv[2], 1);
}
