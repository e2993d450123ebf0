/* IPC transfer records of the custrings API over the MI355X back-end.
 *
 * Same struct names and the same two calls per class as the reference's cpp/include/ipc_transfer.h:31-200
 * (NVStrings::create_ipc_transfer / create_from_ipc, NVCategory likewise), with the native column layout
 * inside: the HIP IPC handles of the chars / offsets / validity buffers and their sizes (custrings_amd.h:
 * cs_ipc_column, cs_ipc_category) instead of an array of custring_view pointers to rebase.  The records are
 * plain bytes: send them to the other process as they are.  The exporting instance must outlive every
 * instance created from its record.
 */
#ifndef NVSTRINGS_AMD_IPC_TRANSFER_H
#define NVSTRINGS_AMD_IPC_TRANSFER_H

#include "custrings_amd.h"

struct nvstrings_ipc_transfer {
  cs_ipc_column column;
};
struct nvcategory_ipc_transfer {
  cs_ipc_category category;
};

#endif
