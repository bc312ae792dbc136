// mock_mx.cpp -- heap implementation of the mx* / mex* subset declared in tests/mock_mex/mex.h (test infrastructure).
//
// Semantics mirrored from MATLAB's documented behaviour, because the gateway relies on them:
//  * arrays are column-major, zero-initialised on creation; mxGetM = first dimension, mxGetN = product of the others;
//  * arrays a MEX function creates and does not return through plhs are destroyed when it returns, and ALL arrays it
//    created are destroyed when it raises an error (the gateway leaks nothing on its error paths by relying on this);
//  * mexErrMsgIdAndTxt does not return.  MATLAB long-jumps; here it throws MockMexError, which skips destructors of
//    nothing the gateway owns only if the gateway keeps its promise of calling it with no C++ object alive -- the
//    harness cannot check that promise (tests/test_mex_gateway_compiles.py checks it on the source), but an exception
//    keeps the process intact either way.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "mex.h"

struct mxArray_tag {
  mxClassID cls = mxDOUBLE_CLASS;
  std::vector<mwSize> dims;
  std::vector<unsigned char> data;          // numeric, logical, char (2 bytes per character)
  std::vector<std::string> fieldnames;      // struct
  std::vector<mxArray*> fields;             // struct: element-major [index * nfields + field]
};

namespace {
struct MockMexError {
  std::string id, msg;
};
std::set<mxArray*> g_live;                  // every array alive
std::vector<mxArray*>* g_call_created = nullptr;   // arrays created inside the running mexFunction
void (*g_at_exit)(void) = nullptr;
int g_lock = 0;

size_t elem_size(mxClassID c) {
  switch (c) {
    case mxDOUBLE_CLASS: case mxINT64_CLASS: case mxUINT64_CLASS: return 8;
    case mxSINGLE_CLASS: case mxINT32_CLASS: case mxUINT32_CLASS: return 4;
    case mxINT16_CLASS: case mxUINT16_CLASS: case mxCHAR_CLASS: return 2;
    case mxINT8_CLASS: case mxUINT8_CLASS: case mxLOGICAL_CLASS: return 1;
    default: return 0;
  }
}
size_t numel(const mxArray* a) {
  size_t n = 1;
  for (mwSize d : a->dims) n *= d;
  return n;
}
mxArray* make(mxClassID c, mwSize ndim, const mwSize* dims) {
  mxArray* a = new mxArray_tag;
  a->cls = c;
  a->dims.assign(dims, dims + ndim);
  while (a->dims.size() < 2) a->dims.push_back(1);
  while (a->dims.size() > 2 && a->dims.back() == 1) a->dims.pop_back();   // MATLAB drops trailing singleton dimensions
  a->data.assign(numel(a) * elem_size(c), 0);
  g_live.insert(a);
  if (g_call_created) g_call_created->push_back(a);
  return a;
}
int field_index(const mxArray* a, const char* name) {
  for (size_t i = 0; i < a->fieldnames.size(); ++i)
    if (a->fieldnames[i] == name) return (int)i;
  return -1;
}
void destroy(mxArray* a) {
  if (!a || !g_live.count(a)) return;
  g_live.erase(a);
  for (mxArray* f : a->fields) destroy(f);
  delete a;
}
}  // namespace

extern "C" {

mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity) { mwSize d[2] = {m, n}; return make(mxDOUBLE_CLASS, 2, d); }
mxArray* mxCreateDoubleScalar(double v) { mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL); *(double*)a->data.data() = v; return a; }
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID c, mxComplexity) { mwSize d[2] = {m, n}; return make(c, 2, d); }
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID c, mxComplexity) { return make(c, ndim, dims); }
mxArray* mxCreateLogicalScalar(bool v) { mwSize d[2] = {1, 1}; mxArray* a = make(mxLOGICAL_CLASS, 2, d); a->data[0] = v ? 1 : 0; return a; }
mxArray* mxCreateString(const char* s) {
  const size_t n = strlen(s);
  mwSize d[2] = {n ? (mwSize)1 : (mwSize)0, (mwSize)n};
  mxArray* a = make(mxCHAR_CLASS, 2, d);
  for (size_t i = 0; i < n; ++i) ((mxChar*)a->data.data())[i] = (unsigned char)s[i];
  return a;
}
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** names) {
  mwSize d[2] = {m, n};
  mxArray* a = make(mxSTRUCT_CLASS, 2, d);
  for (int i = 0; i < nfields; ++i) a->fieldnames.push_back(names[i]);
  a->fields.assign(m * n * (size_t)nfields, nullptr);
  return a;
}
void mxDestroyArray(mxArray* a) { destroy(a); }

mxClassID mxGetClassID(const mxArray* a) { return a->cls; }
mxDouble* mxGetDoubles(const mxArray* a) {
  if (a->cls != mxDOUBLE_CLASS) throw MockMexError{"mock:mxGetDoubles", "mxGetDoubles on a non-double array"};   // MATLAB returns NULL
  return a->data.empty() ? nullptr : (mxDouble*)a->data.data();
}
void* mxGetData(const mxArray* a) { return a->data.empty() ? nullptr : (void*)a->data.data(); }
double mxGetScalar(const mxArray* a) {
  if (a->data.empty()) throw MockMexError{"mock:mxGetScalar", "mxGetScalar on an empty array"};   // undefined in MATLAB
  const void* p = a->data.data();
  switch (a->cls) {
    case mxDOUBLE_CLASS: return *(const double*)p;
    case mxSINGLE_CLASS: return *(const float*)p;
    case mxINT64_CLASS: return (double)*(const int64_t*)p;
    case mxUINT64_CLASS: return (double)*(const uint64_t*)p;
    case mxINT32_CLASS: return *(const int32_t*)p;
    case mxUINT32_CLASS: return *(const uint32_t*)p;
    case mxINT16_CLASS: return *(const int16_t*)p;
    case mxUINT16_CLASS: case mxCHAR_CLASS: return *(const uint16_t*)p;
    case mxINT8_CLASS: return *(const int8_t*)p;
    case mxUINT8_CLASS: case mxLOGICAL_CLASS: return *(const uint8_t*)p;
    default: throw MockMexError{"mock:mxGetScalar", "mxGetScalar on a struct / cell"};
  }
}
size_t mxGetM(const mxArray* a) { return a->dims[0]; }
size_t mxGetN(const mxArray* a) { size_t n = 1; for (size_t i = 1; i < a->dims.size(); ++i) n *= a->dims[i]; return n; }
size_t mxGetNumberOfElements(const mxArray* a) { return numel(a); }
size_t mxGetElementSize(const mxArray* a) { return elem_size(a->cls); }
mwSize mxGetNumberOfDimensions(const mxArray* a) { return a->dims.size(); }
const mwSize* mxGetDimensions(const mxArray* a) { return a->dims.data(); }
mxArray* mxGetField(const mxArray* a, mwIndex index, const char* name) {
  if (a->cls != mxSTRUCT_CLASS || index >= numel(a)) return nullptr;
  const int f = field_index(a, name);
  return f < 0 ? nullptr : a->fields[index * a->fieldnames.size() + f];
}
void mxSetField(mxArray* a, mwIndex index, const char* name, mxArray* value) {
  const int f = field_index(a, name);
  if (a->cls != mxSTRUCT_CLASS || f < 0 || index >= numel(a)) throw MockMexError{"mock:mxSetField", "no such field"};
  a->fields[index * a->fieldnames.size() + f] = value;
  if (g_call_created)      // owned by the struct from now on
    for (auto& c : *g_call_created)
      if (c == value) c = nullptr;
}
int mxGetNumberOfFields(const mxArray* a) { return (int)a->fieldnames.size(); }
const char* mxGetFieldNameByNumber(const mxArray* a, int i) { return (i >= 0 && i < (int)a->fieldnames.size()) ? a->fieldnames[i].c_str() : nullptr; }
int mxGetString(const mxArray* a, char* buf, mwSize buflen) {
  if (a->cls != mxCHAR_CLASS || buflen == 0) return 1;
  const size_t n = numel(a);
  const size_t m = n < buflen - 1 ? n : buflen - 1;
  for (size_t i = 0; i < m; ++i) buf[i] = (char)((const mxChar*)a->data.data())[i];
  buf[m] = 0;
  return n > buflen - 1 ? 1 : 0;    // 1 = truncated, like MATLAB
}
bool mxIsChar(const mxArray* a) { return a->cls == mxCHAR_CLASS; }
bool mxIsDouble(const mxArray* a) { return a->cls == mxDOUBLE_CLASS; }
bool mxIsStruct(const mxArray* a) { return a->cls == mxSTRUCT_CLASS; }
bool mxIsEmpty(const mxArray* a) { return numel(a) == 0; }
bool mxIsLogicalScalarTrue(const mxArray* a) { return a->cls == mxLOGICAL_CLASS && numel(a) == 1 && a->data[0] != 0; }

void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw MockMexError{id ? id : "", buf};
}
void mexErrMsgTxt(const char* msg) { throw MockMexError{"", msg ? msg : ""}; }
void mexLock(void) { ++g_lock; }
void mexUnlock(void) { if (g_lock > 0) --g_lock; }
int mexAtExit(void (*fn)(void)) { g_at_exit = fn; return 0; }

// One MATLAB-style invocation: returns 0 on success, 1 if the MEX function raised (id / message copied out).
int mock_mex_call(int nlhs, mxArray** plhs, int nrhs, const mxArray** prhs, char* errid, size_t idlen, char* errmsg, size_t msglen) {
  std::vector<mxArray*> created;
  g_call_created = &created;
  const int nout = nlhs > 1 ? nlhs : 1;       // MATLAB always provides plhs[0] (ans)
  for (int i = 0; i < nout; ++i) plhs[i] = nullptr;
  int rc = 0;
  try {
    mexFunction(nlhs, plhs, nrhs, prhs);
  } catch (const MockMexError& e) {
    snprintf(errid, idlen, "%s", e.id.c_str());
    snprintf(errmsg, msglen, "%s", e.msg.c_str());
    rc = 1;
  } catch (const std::exception& e) {
    snprintf(errid, idlen, "mock:exception");
    snprintf(errmsg, msglen, "%s", e.what());
    rc = 1;
  }
  g_call_created = nullptr;
  std::set<mxArray*> keep;
  if (rc == 0)
    for (int i = 0; i < nout; ++i)
      if (plhs[i]) keep.insert(plhs[i]);
  for (mxArray* a : created)
    if (a && !keep.count(a)) destroy(a);          // temporaries (or, after an error, everything the call created)
  if (rc)
    for (int i = 0; i < nout; ++i) plhs[i] = nullptr;
  return rc;
}
void mock_mex_run_at_exit(void) { if (g_at_exit) g_at_exit(); }
long mock_mex_live_arrays(void) { return (long)g_live.size(); }
int mock_mex_lock_count(void) { return g_lock; }

}  // extern "C"
