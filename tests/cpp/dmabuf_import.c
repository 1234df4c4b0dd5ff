/* The NON-HIP importing side of tests/test_gpu_mesh_export.py: a stand-in for the renderer's Vulkan / wgpu import of the
 * library's mesh buffers. It takes a dma-buf file descriptor (inherited from the exporting process), imports it as a buffer object through
 * libdrm_amdgpu — the kernel graphics driver's user-space API, nothing of HIP / ROCr in this program —, maps it and writes
 * [offset, offset + size) of it to a file for the test to compare with ivx_mesh_download.
 * usage: dmabuf_import <fd> <offset> <size> <out-file>      exit 0 = bytes written; otherwise the failing call is on stderr. */
#include <amdgpu.h>
#include <amdgpu_drm.h>
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

static int try_node(const char* node, int dmabuf, uint64_t offset, uint64_t size, const char* out_path) {
    int drm = open(node, O_RDWR | O_CLOEXEC);
    if (drm < 0) {
        fprintf(stderr, "%s: open: %s\n", node, strerror(errno));
        return 1;
    }
    uint32_t major = 0, minor = 0;
    amdgpu_device_handle dev;
    int rc = amdgpu_device_initialize(drm, &major, &minor, &dev);
    if (rc) {
        fprintf(stderr, "%s: amdgpu_device_initialize: %s\n", node, strerror(-rc));
        close(drm);
        return 1;
    }
    struct amdgpu_bo_import_result imp;
    memset(&imp, 0, sizeof(imp));
    rc = amdgpu_bo_import(dev, amdgpu_bo_handle_type_dma_buf_fd, (uint32_t)dmabuf, &imp);
    if (rc) {
        fprintf(stderr, "%s: amdgpu_bo_import(dma_buf_fd %d): %s\n", node, dmabuf, strerror(-rc));
        amdgpu_device_deinitialize(dev);
        close(drm);
        return 1;
    }
    int ok = 1;
    if (offset + size > imp.alloc_size) {
        fprintf(stderr, "%s: the imported object has %llu bytes, [%llu, %llu) asked for\n", node, (unsigned long long)imp.alloc_size,
                (unsigned long long)offset, (unsigned long long)(offset + size));
        ok = 0;
    }
    void* cpu = NULL;
    if (ok) {
        rc = amdgpu_bo_cpu_map(imp.buf_handle, &cpu);
        if (rc) {
            fprintf(stderr, "%s: amdgpu_bo_cpu_map: %s\n", node, strerror(-rc));
            ok = 0;
        }
    }
    if (ok) {
        FILE* f = fopen(out_path, "wb");
        if (!f || fwrite((const uint8_t*)cpu + offset, 1, (size_t)size, f) != (size_t)size) {
            fprintf(stderr, "writing %s failed\n", out_path);
            ok = 0;
        }
        if (f) fclose(f);
        amdgpu_bo_cpu_unmap(imp.buf_handle);
    }
    if (ok) fprintf(stderr, "%s: imported %llu bytes of a %llu-byte buffer object (drm %u.%u)\n", node, (unsigned long long)size,
                    (unsigned long long)imp.alloc_size, major, minor);
    amdgpu_bo_free(imp.buf_handle);
    amdgpu_device_deinitialize(dev);
    close(drm);
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: dmabuf_import <fd> <offset> <size> <out-file> [render node]\n");
        return 2;
    }
    const int dmabuf = atoi(argv[1]);
    const uint64_t offset = strtoull(argv[2], NULL, 10), size = strtoull(argv[3], NULL, 10);
    if (argc > 5) return try_node(argv[5], dmabuf, offset, size, argv[4]);
    /* the render node of the exporting GPU is not known here: the object imports (and maps) on its own device */
    for (int n = 128; n < 192; ++n) {
        char node[64];
        snprintf(node, sizeof(node), "/dev/dri/renderD%d", n);
        if (access(node, F_OK) != 0) continue;
        if (try_node(node, dmabuf, offset, size, argv[4]) == 0) return 0;
    }
    fprintf(stderr, "no render node under /dev/dri imported and mapped the descriptor\n");
    return 1;
}
