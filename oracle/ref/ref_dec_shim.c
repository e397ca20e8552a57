/*
 * ref_dec_shim.c -- TEST INFRASTRUCTURE ONLY (oracle/_ref/libnhwref_dec.so).
 *
 * Links the UNMODIFIED reference decoder sources where they lie under /root/reference/decoder
 * (never copied into this repo) with
 *   1. the canonical allocation model (zero fill + 4 KiB zero guards on both sides, as ref_shim.c),
 *      so any out-of-bounds or never-written read of the reference returns 0;
 *   2. nhwref_decode_planes(): decode_image() (nhw_decoder.c:54) on a file, returning the three
 *      512x512 planes im_bufferY/U/V it leaves behind -- the checkpoint before the colour matrix;
 *   3. nhwref_decode_bmp(): the reference CLI main (nhw_decoder_cli.c:70, renamed at compile time
 *      with -Dmain=nhwref_dec_cli_main) writing the BMP it would write;
 *   4. exit() capture -> status code.
 * Nothing in the product may link or load this.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <setjmp.h>
#include "codec.h" /* the reference's own header, via -I/root/reference/decoder */

#define GUARD 4096
#define MAX_LIVE 8192
extern void *__real_calloc(size_t, size_t);
extern void __real_free(void *);
static void *g_live[MAX_LIVE];
static int g_nlive = 0, g_track = 0;

static void *guard_alloc(size_t n)
{
	unsigned char *p = (unsigned char *)__real_calloc(1, n + 2 * GUARD);
	if (!p) return NULL;
	p += GUARD;
	if (g_track && g_nlive < MAX_LIVE) g_live[g_nlive++] = p;
	return p;
}
void *__wrap_malloc(size_t n) { return guard_alloc(n); }
void *__wrap_calloc(size_t a, size_t b) { return guard_alloc(a * b); }
void __wrap_free(void *p)
{
	int i;
	if (!p) return;
	for (i = g_nlive - 1; i >= 0; i--)
		if (g_live[i] == p) { g_live[i] = g_live[--g_nlive]; break; }
	__real_free((unsigned char *)p - GUARD);
}
static void release_leftovers(void)
{
	while (g_nlive > 0) __real_free((unsigned char *)g_live[--g_nlive] - GUARD);
}

static jmp_buf g_jmp;
static int g_in_call = 0;
extern void __real_exit(int);
void __wrap_exit(int code)
{
	if (g_in_call) longjmp(g_jmp, code ? code : 1000);
	__real_exit(code);
}

extern int nhwref_dec_cli_main(int argc, char **argv);

/* planes: 3 x 262144 bytes (Y, U, V as decode_image leaves them); returns 0, or the exit() code */
int nhwref_decode_planes(const char *nhw_path, uint8_t *planes, int *quality)
{
	image_buffer im;
	decode_state dec;
	int rc;
	memset(&im, 0, sizeof im); memset(&dec, 0, sizeof dec);
	g_track = 1; g_in_call = 1;
	rc = setjmp(g_jmp);
	if (!rc) {
		decode_image(&im, &dec, (char *)nhw_path);
		memcpy(planes, im.im_bufferY, 4 * IM_SIZE);
		memcpy(planes + 4 * IM_SIZE, im.im_bufferU, 4 * IM_SIZE);
		memcpy(planes + 8 * IM_SIZE, im.im_bufferV, 4 * IM_SIZE);
		if (quality) *quality = im.setup->quality_setting;
	}
	g_in_call = 0; g_track = 0;
	release_leftovers();
	return rc;
}

int nhwref_decode_bmp(const char *nhw_path, const char *bmp_path)
{
	char a0[] = "nhw-dec";
	char *argv[4];
	int rc;
	argv[0] = a0; argv[1] = (char *)nhw_path; argv[2] = (char *)bmp_path; argv[3] = NULL;
	g_track = 1; g_in_call = 1;
	rc = setjmp(g_jmp);
	if (!rc) rc = nhwref_dec_cli_main(3, argv);
	g_in_call = 0; g_track = 0;
	release_leftovers();
	return rc;
}
