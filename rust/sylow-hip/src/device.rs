//! Device plumbing over the C ABI: error type, one context per GPU, RAII device buffers, uploads in the engine's
//! struct-of-arrays layout.  Nothing here knows about curves.
use crate::ffi;
use std::ffi::CStr;
use std::marker::PhantomData;
use std::os::raw::c_void;
use std::ptr;

/// A failed call into libsylow_hip.so (negative SYLOW_HIP_E_* code + the library's message).
#[derive(Debug, Clone)]
pub struct Error {
    pub code: i32,
    pub message: String,
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "sylow_hip error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}

pub(crate) fn check(code: i32) -> Result<(), Error> {
    if code == 0 {
        return Ok(());
    }
    // SAFETY: sylow_hip_last_error returns a pointer to a thread-local NUL-terminated buffer owned by the library.
    let message = unsafe { CStr::from_ptr(ffi::sylow_hip_last_error()) }.to_string_lossy().into_owned();
    Err(Error { code, message })
}

/// One GPU (one process per GPU is the intended deployment; `ordinal` = LOCAL_RANK).  `stream` is a raw hipStream_t,
/// null = the default stream.  Every call re-asserts the device for the calling thread (sylow_hip_set_device).
pub struct Device {
    pub ordinal: i32,
    pub stream: *mut c_void,
}

// The library's entry points may be called from several host threads (include/sylow_hip.h, "Threads").
unsafe impl Send for Device {}
unsafe impl Sync for Device {}

impl Device {
    pub fn new(ordinal: i32) -> Result<Self, Error> {
        // SAFETY: plain value arguments.
        check(unsafe { ffi::sylow_hip_init(ordinal) })?;
        Ok(Device { ordinal, stream: ptr::null_mut() })
    }

    pub fn with_stream(ordinal: i32, stream: *mut c_void) -> Result<Self, Error> {
        let mut d = Device::new(ordinal)?;
        d.stream = stream;
        Ok(d)
    }

    pub(crate) fn bind(&self) -> Result<(), Error> {
        // SAFETY: plain value argument.
        check(unsafe { ffi::sylow_hip_set_device(self.ordinal) })
    }

    pub fn sync(&self) -> Result<(), Error> {
        self.bind()?;
        // SAFETY: `stream` is null or a live hipStream_t supplied by the caller.
        check(unsafe { ffi::sylow_hip_stream_sync(self.stream) })
    }

    /// Uninitialised device buffer of `len` elements.
    pub fn alloc<T: Copy>(&self, len: usize) -> Result<DeviceBuf<T>, Error> {
        self.bind()?;
        let mut p: *mut c_void = ptr::null_mut();
        // SAFETY: `p` is a valid out-pointer.
        check(unsafe { ffi::sylow_hip_malloc(&mut p, len * std::mem::size_of::<T>()) })?;
        Ok(DeviceBuf { ptr: p, len, _t: PhantomData })
    }

    /// Host slice -> device, byte for byte.
    pub fn upload<T: Copy>(&self, host: &[T]) -> Result<DeviceBuf<T>, Error> {
        let buf = self.alloc::<T>(host.len())?;
        // SAFETY: both ranges are `host.len() * size_of::<T>()` bytes long.
        check(unsafe {
            ffi::sylow_hip_memcpy_h2d(buf.ptr, host.as_ptr() as *const c_void, std::mem::size_of_val(host), self.stream)
        })?;
        self.sync()?;
        Ok(buf)
    }

    /// Array-of-structs host data (`n` objects of `W` words each, e.g. `Vec<[u64; 8]>` for affine G1 points) -> the engine's
    /// struct-of-arrays layout `[W][n]`, transposed on the device.
    pub fn upload_soa<const W: usize>(&self, aos: &[[u64; W]]) -> Result<DeviceBuf<u64>, Error> {
        let n = aos.len();
        let staged = self.alloc::<u64>(W * n)?;
        let soa = self.alloc::<u64>(W * n)?;
        // SAFETY: `aos` is `W * n` contiguous u64; both device buffers hold `W * n` words.
        check(unsafe { ffi::sylow_hip_memcpy_h2d(staged.ptr, aos.as_ptr() as *const c_void, W * n * 8, self.stream) })?;
        check(unsafe { ffi::sylow_hip_aos_to_soa(staged.as_ptr(), soa.as_mut_ptr(), W, n, self.stream) })?;
        self.sync()?;
        Ok(soa)
    }

    /// Device struct-of-arrays `[W][n]` -> host array-of-structs.
    pub fn download_aos<const W: usize>(&self, soa: &DeviceBuf<u64>, n: usize) -> Result<Vec<[u64; W]>, Error> {
        assert_eq!(soa.len, W * n);
        let staged = self.alloc::<u64>(W * n)?;
        let mut host = vec![[0u64; W]; n];
        // SAFETY: buffers hold `W * n` words each.
        check(unsafe { ffi::sylow_hip_soa_to_aos(soa.as_ptr(), staged.as_mut_ptr(), W, n, self.stream) })?;
        check(unsafe { ffi::sylow_hip_memcpy_d2h(host.as_mut_ptr() as *mut c_void, staged.ptr as *const c_void, W * n * 8, self.stream) })?;
        self.sync()?;
        Ok(host)
    }

    pub fn download<T: Copy + Default>(&self, buf: &DeviceBuf<T>) -> Result<Vec<T>, Error> {
        let mut host = vec![T::default(); buf.len];
        // SAFETY: `host` and `buf` are `buf.len` elements long.
        check(unsafe {
            ffi::sylow_hip_memcpy_d2h(host.as_mut_ptr() as *mut c_void, buf.ptr as *const c_void, buf.len * std::mem::size_of::<T>(), self.stream)
        })?;
        self.sync()?;
        Ok(host)
    }
}

/// hipMalloc'ed memory, freed on drop.
pub struct DeviceBuf<T> {
    ptr: *mut c_void,
    pub len: usize,
    _t: PhantomData<T>,
}

impl<T> DeviceBuf<T> {
    pub fn as_ptr(&self) -> *const T {
        self.ptr as *const T
    }
    pub fn as_mut_ptr(&self) -> *mut T {
        self.ptr as *mut T
    }
}

impl<T> Drop for DeviceBuf<T> {
    fn drop(&mut self) {
        // SAFETY: `ptr` came from sylow_hip_malloc and is freed once (hipFree synchronises with pending work).
        unsafe { ffi::sylow_hip_free(self.ptr) };
    }
}
